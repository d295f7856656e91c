#!/usr/bin/env python3
"""HBM traffic of k_solve's phases one by one: the probes below are run under
  rocprofv3 --pmc FETCH_SIZE   and   rocprofv3 --pmc WRITE_SIZE   (separate passes, counters only)
and `phase_traffic.py --summarise DIR` turns the per-dispatch counters into measured / needed bytes per phase.

Dispatches of one run (512 QPs = one per resident workgroup, n = 1000, full-chip contention):
  k_sweep_probe #1  reps 0            form_schur + dense_factor only            -> "factor"
  k_sweep_probe #2  reps R, 16 ranks  the same + 2R sweeps of 16 ranks          -> (#2 - #1) / 2R = one 16-rank sweep
  k_sweep_probe #3  reps R,  8 ranks  the same + 2R sweeps of  8 ranks          -> (#3 - #1) / 2R = one 8-rank sweep
  k_ldlsolve_all    reps S            S triangular solves per QP                -> / S = one solve
Needed bytes: sweep = 16 B x entries swept (device counter); solve = 2 nnz(L) 8 + 24 n; factor = nnz(L) 8 written + the panel
re-reads of the left-looking update (device counter) + the assembly's reads of A, Q.
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(args):
    import numpy as np
    import torch
    torch.cuda.init()
    from qpalm_amd.problems import random_qp
    from qpalm_amd.solver import Context, QpalmBatch
    ctx = Context(0, lib_path=args.lib)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    n, m, B = args.n, 2 * args.n, args.batch
    base = [random_qp(n, m, seed=1000 + k, density_A=0.01, density_M=0.005) for k in range(8)]
    probs = [base[k % 8] for k in range(B)]
    bt = QpalmBatch(ctx, probs, ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
    bt.warm_start(None, None)
    out = {"n": n, "m": m, "batch": B, "reps": args.reps, "solve_reps": args.solve_reps}
    ms0 = bt.sweep_probe(0, 16)
    st = bt.stats_all()
    out["factor"] = {"ms": ms0, "reread_entries": float(np.mean([s.factor_reread_entries for s in st]))}
    for ranks in (16, 8, 64):   # 64: four sweeps per call of the update (per-call costs -- callee-saved registers to scratch -- spread over four)
        ms = bt.sweep_probe(args.reps, ranks)
        st = bt.stats_all()
        out["sweep%d" % ranks] = {"ms": ms, "sweeps": float(np.mean([s.n_sweeps for s in st])),
                                  "entries": float(np.mean([s.sweep_entries for s in st]))}
    out["solve"] = {"ms_per_rep": bt.ldlsolve_all(args.solve_reps)}
    out["nnzA"] = int(probs[0].Ap[-1]); out["nnzQ"] = int(probs[0].Qp[-1])
    print(json.dumps(out))


def per_dispatch(d, counter):
    """[(kernel name, value)] in dispatch order"""
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows.append((int(r.get("Dispatch_Id", 0)), r.get("Kernel_Name", ""), float(r["Counter_Value"])))
    rows.sort()
    return [(k, v) for _, k, v in rows]


def summarise(d):
    info = None
    for name in ("fetch.json", "write.json"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            for ln in open(p):
                if ln.startswith("{"):
                    info = json.loads(ln)
    fe = per_dispatch(os.path.join(d, "pmc_fetch"), "FETCH_SIZE")
    wr = per_dispatch(os.path.join(d, "pmc_write"), "WRITE_SIZE")

    def pick(rows, key):
        return [v for k, v in rows if key in k]
    res = {"run": info, "unit": "MB per QP", "gfx950_correction": "FETCH_SIZE x 2 (KB), WRITE_SIZE as reported (KB)"}
    if not info:
        print(json.dumps({"error": "no run record"}))
        return
    B, n = info["batch"], info["n"]
    nnzL = n * (n + 1) // 2
    R2 = 2 * info["reps"]
    fs, ws = pick(fe, "k_sweep_probe"), pick(wr, "k_sweep_probe")
    fl, wl = pick(fe, "k_ldlsolve_all"), pick(wr, "k_ldlsolve_all")
    KB = 1024.0 / 1e6 / B   # counter (KB per launch) -> MB per QP
    if len(fs) >= 3 and len(ws) >= 3:
        need_factor = (nnzL * 8 + info["factor"]["reread_entries"] * 8 + (info["nnzA"] + info["nnzQ"]) * 12) / 1e6
        res["factor"] = {"read": 2 * fs[0] * KB, "write": ws[0] * KB, "needed_write": nnzL * 8 / 1e6,
                         "needed_incl_documented_rereads": need_factor, "compulsory": (nnzL * 8 + (info["nnzA"] + info["nnzQ"]) * 12) / 1e6}
        for j, key in ((1, "sweep16"), (2, "sweep8"), (3, "sweep64")):
            if key not in info or len(fs) <= j:
                continue
            nsw = max(info[key]["sweeps"], 1)    # sweeps of the probe launch (2 R calls x ceil(ranks / 16))
            ent = info[key]["entries"] / nsw
            res[key] = {"read": 2 * (fs[j] - fs[0]) * KB / nsw, "write": (ws[j] - ws[0]) * KB / nsw, "needed_read": ent * 8 / 1e6, "needed_write": ent * 8 / 1e6,
                        "us_per_sweep": 1e3 * (info[key]["ms"] - info["factor"]["ms"]) / nsw, "sweeps_per_call": nsw / R2}
    if fl and wl:
        S = info["solve_reps"]
        res["solve"] = {"read": 2 * fl[0] * KB / S, "write": wl[0] * KB / S, "needed_read": (2 * nnzL * 8 + 16 * n) / 1e6, "needed_write": 8 * n / 1e6,
                        "ms_per_solve": info["solve"]["ms_per_rep"]}
    for k in ("factor", "sweep16", "sweep8", "sweep64", "solve"):
        if k in res:
            r = res[k]
            need = r.get("needed_incl_documented_rereads", r.get("needed_read", 0) + r.get("needed_write", 0))
            r["moved"] = r["read"] + r["write"]
            r["moved_over_needed"] = r["moved"] / need if need else None
    print(json.dumps(res, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--solve-reps", type=int, default=8)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--opt", action="append", default=[], help="context option name=value (e.g. ld_align=8)")
    ap.add_argument("--summarise", default=None)
    args = ap.parse_args()
    if args.summarise:
        return summarise(args.summarise)
    return run(args)


if __name__ == "__main__":
    main()
