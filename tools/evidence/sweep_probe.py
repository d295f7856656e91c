#!/usr/bin/env python3
"""The rank-update sweep alone under full-chip contention (qpg_batch_sweep_probe), for same-box comparisons of build variants.
usage: sweep_probe.py [--n 1000] [--batch 512] [--reps 4] [--ranks 16] name [name ...]
(name = suffix of qpalm_amd/lib/libqpalm_gfx950_<name>.so, "cur" = the shipped build; each variant runs in its own process)"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(args, lib):
    import numpy as np
    import torch
    torch.cuda.init()
    from qpalm_amd.problems import random_qp
    from qpalm_amd.solver import Context, QpalmBatch
    ctx = Context(0, lib_path=lib)
    n, m = args.n, 2 * args.n
    base = [random_qp(n, m, seed=1000 + k, density_A=0.01 if n >= 400 else 4.0 / n, density_M=0.005 if n >= 400 else 2.0 / n) for k in range(8)]
    probs = [base[k % 8] for k in range(args.batch)]
    bt = QpalmBatch(ctx, probs, ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
    bt.warm_start(None, None)
    out = []
    for ranks in args.ranks:
        bt.sweep_probe(1, ranks)
        ms = bt.sweep_probe(args.reps, ranks)
        st = bt.stats_all()
        mean = lambda f: float(np.mean([f(s) for s in st]))
        nsw = mean(lambda s: s.n_sweeps)
        g = [mean(lambda s, k=k: s.ms_dbg[k]) for k in range(16)]
        out.append(dict(ranks=ranks, launch_ms=ms, sweeps=nsw, us_per_sweep=1e3 * mean(lambda s: s.ms_update) / max(nsw, 1),
                        panel_us=1e3 * g[1] / max(nsw, 1), trail_us=1e3 * g[2] / max(nsw, 1), wall_us=1e3 * g[7] / max(nsw, 1),
                        dbg8_11_us=[round(1e3 * g[k] / max(nsw, 1), 1) for k in (8, 9, 10, 11)], entries=mean(lambda s: s.sweep_entries) / max(nsw, 1)))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--ranks", type=int, nargs="+", default=[16])
    ap.add_argument("--one", default=None)
    ap.add_argument("names", nargs="*")
    args = ap.parse_args()
    if args.one is not None:
        return one(args, args.one)
    for name in args.names:
        lib = os.path.join(ROOT, "qpalm_amd", "lib", "libqpalm_gfx950.so" if name == "cur" else "libqpalm_gfx950_%s.so" % name)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--n", str(args.n), "--batch", str(args.batch), "--reps", str(args.reps),
                            "--ranks"] + [str(x) for x in args.ranks] + ["--one", lib], capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("[")]
        if r.returncode != 0 or not line:
            print(name, "FAILED", r.stderr[-400:])
            continue
        for rec in json.loads(line[-1]):
            print("%-8s ranks %2d: %7.1f us/sweep (sweep wall %7.1f)  panel %7.1f  trailing %7.1f  dbg[8..11] %s  launch %.2f ms  sweeps %.0f entries %.0f" % (
                name, rec["ranks"], rec["us_per_sweep"], rec["wall_us"], rec["panel_us"], rec["trail_us"], rec["dbg8_11_us"], rec["launch_ms"], rec["sweeps"], rec["entries"]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
