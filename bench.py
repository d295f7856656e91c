#!/usr/bin/env python3
"""bench.py -- QP solves/sec on batched random QPs (n=1000, m=2000, ~1 % dense A), BASELINE.json's metric.

One "step" = one pass of the hot path over one batch: every QP of the (HBM-resident, already
scaled) batch is cold-started (qpalm_warm_start(NULL, NULL)) and solved to eps 1e-6 by the persistent
gfx950 kernel; with N > 1 ranks every rank owns its own shard of B QPs (weak scaling) and the
solutions are gathered to rank 0 over RCCL inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_solve): algorithmic bytes of
SURVEY.md section 8d, counted from the device-side work counters, divided by the kernel's HIP-event
duration.  `cpu_baseline` times the CPU oracle ("port") on a bounded sample of the same QPs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy ceiling)
PMC_FILE = os.path.join(ROOT, "profiles", "r01", "k_solve_pmc_traffic.json")


def pmc_traffic(batch, n, m):
    """HBM bytes per k_solve launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, collected in
    separate runs of this very command, corrected as MI355X_MICROARCH.md prescribes); None when no pass matches."""
    try:
        with open(PMC_FILE) as f:
            d = json.load(f)
        if d["batch"] == batch and d["n"] == n and d["m"] == m:
            return float(d["traffic_bytes_per_launch"])
    except Exception:
        pass
    return None



def algorithmic_bytes(n, m, nnzA, nnzQ, stats, iters):
    """SURVEY.md section 8d byte model, per QP (fp64 = 8 B, int32 indices = 4 B)."""
    nnzL = n * (n + 1) // 2
    b_solve = 2 * nnzL * 8 + 8 * n + 16 * n
    b_spmv_A = nnzA * 12 + 4 * (n + 1) + 8 * (m + n)
    b_spmv_Q = nnzQ * 12 + 4 * (n + 1) + 16 * n
    b_vec = 8 * (22 * m + 18 * n) + 2 * (2 * m * 12)
    b_newton = b_solve + 2 * b_spmv_A + b_spmv_Q + b_vec
    b_outer = b_spmv_A + b_vec
    b_refactor = nnzL * 8 + (nnzQ + nnzA) * 12
    b_sweep = 2 * 8 * nnzL  # upper bound of section 8d: a sweep touches L[:, j0:] once, read + write
    n_newton = stats["n_solve"]
    total = n_newton * b_newton + max(iters - n_newton, 0) * b_outer
    total += (stats["n_refactor"] + stats["n_factor_Q"]) * b_refactor + stats["n_sweeps"] * b_sweep
    return total, dict(b_solve=b_solve, b_newton=b_newton, b_sweep=b_sweep, b_refactor=b_refactor)


def cpu_baseline(problems, settings_kw, budget_s=20.0):
    """Oracle ("port") on host cores, one QP per thread; bounded sample."""
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as ob
    cores = max(1, min(os.cpu_count() or 1, 64))
    libpath = None
    try:  # host-tuned build of the same source (the GPU box's CPU may differ from the build host)
        out = os.path.join("/tmp", "libqpalm_oracle_native_%d.so" % os.getpid())
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native", "OUT=" + out],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        libpath = out
    except Exception:
        libpath = None

    def one(p):
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**settings_kw), libpath=libpath)
        t0 = time.perf_counter()
        o.solve()
        dt = time.perf_counter() - t0
        st = o.status_val
        o.cleanup()
        return dt, st

    t_probe, _ = one(problems[0])
    nsample = int(max(cores, min(len(problems), cores * max(1.0, budget_s / max(t_probe, 1e-3)) / 1.0)))
    nsample = min(nsample, len(problems), 4 * cores)
    sample = problems[:nsample]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(one, sample))
    wall = time.perf_counter() - t0
    assert all(s == 1 for _, s in res)
    return {"value": len(sample) / wall, "unit": "QP/s", "cores": cores, "kind": "port",
            "sample": "%d of the batch's random-1000 QPs, solve phase only (eps 1e-6), one QP per thread, "
                      "oracle/qpalm_oracle.c built -O3 -march=native; single-QP time %.3f s" % (len(sample), t_probe)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("QPALM_BENCH_BATCH", "4096")),
                    help="QPs per GPU (512 resident factor slots = workgroups; the rest queue up behind them)")
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--m", type=int, default=2000)
    ap.add_argument("--rank-threshold", type=int, default=int(os.environ.get("QPALM_RANK_THRESHOLD", "-1")))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--max-slots", type=int, default=0, help="resident factor slots = concurrent workgroups (0: library default)")
    ap.add_argument("--dbg-flags", type=int, default=0, help="timing experiments only (results wrong)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)

    from qpalm_amd.problems import random_qp
    from qpalm_amd.solver import Context, QpalmBatch
    ctx = Context(local)
    ctx.set_option("update_rank_threshold", args.rank_threshold)
    if args.dbg_flags:
        ctx.set_option("dbg_flags", args.dbg_flags)
    if args.max_slots:
        ctx.set_option("max_slots", args.max_slots)
    B, n, m = args.batch, args.n, args.m
    settings_kw = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    dens_A = 0.01 if n >= 400 else max(0.01, 4.0 / n)
    dens_M = 0.005 if n >= 400 else max(0.005, 2.0 / n)
    probs = [random_qp(n, m, seed=1000 + rank * B + k, density_A=dens_A, density_M=dens_M) for k in range(B)]
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**settings_kw))   # upload + Ruiz scaling: not timed

    def gather_solutions():
        if world == 1:
            return
        from qpalm_amd.dist import device_view
        tx = device_view(bt, "solution_x", (B, n), "cuda:%d" % local)   # zero-copy views of the HBM arrays
        ty = device_view(bt, "solution_y", (B, m), "cuda:%d" % local)
        gx = [torch.empty_like(tx) for _ in range(world)] if rank == 0 else None
        gy = [torch.empty_like(ty) for _ in range(world)] if rank == 0 else None
        dist.gather(tx, gx, dst=0)
        dist.gather(ty, gy, dst=0)

    def step():
        bt.warm_start(None, None)
        bt.solve()
        gather_solutions()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        step()
        kernel_ms.append(bt.last_solve_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda:%d" % local, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # parity summary + work counters of the last step
    statuses = bt.statuses()
    iters = np.array([int(bt.info(b).iter) for b in range(B)])
    stats = [bt.stats(b) for b in range(B)]
    keys = ("n_refactor", "n_factor_Q", "n_sweeps", "n_rank1", "n_solve")
    tot_bytes = 0
    for b in range(B):
        sd = {k: int(getattr(stats[b], k)) for k in keys}
        tb, parts = algorithmic_bytes(n, m, int(probs[b].Ap[-1]), int(probs[b].Qp[-1]), sd, int(iters[b]))
        tot_bytes += tb
    kms = float(np.mean(kernel_ms))
    achieved = tot_bytes / (kms * 1e-3) / 1e9
    phase = {k: float(np.mean([getattr(s, k) for s in stats])) for k in ("ms_total", "ms_factor", "ms_update", "ms_solve", "ms_linesearch")}
    phase["dbg"] = [float(np.mean([s.ms_dbg[k] for s in stats])) for k in range(16)]

    out = None
    if rank == 0:
        # stand-alone LDL^T solve kernel (the "HBM GB/s on LDL" half of the metric)
        nsl = min(B, 512)
        ms_ldl = bt.ldlsolve_all(reps=4)
        ldl_bytes = nsl * parts["b_solve"]
        out = {
            "metric": "QP solves/sec (batched random n=%d,m=%d)" % (n, m),
            "value": world * B * args.steps / elapsed, "unit": "QP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "random-%d: batch of %d QPs per GPU, n=%d m=%d nnz(A)~%d nnz(tril Q)~%d, eps 1e-6, "
                                   "scaling 10, cold start" % (n, B, n, m, int(probs[0].Ap[-1]), int(probs[0].Qp[-1])),
                       "batch_per_gpu": B, "parallelism": "batch-shard x%d" % world,
                       "update_rank_threshold": args.rank_threshold},
            "roofline": {"bound": "hbm", "kernel": "k_solve (persistent, one workgroup per QP)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(B, n, m) if world == 1 else None, "kernel_ms": kms, "algorithmic_bytes_per_launch": tot_bytes},
            "ldl_solve": {"kernel": "k_ldlsolve_all", "qps": nsl, "ms": ms_ldl, "bytes": ldl_bytes,
                          "achieved": ldl_bytes / (ms_ldl * 1e-3) / 1e9, "unit": "GB/s",
                          "frac": ldl_bytes / (ms_ldl * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "solve_stats": {"all_solved": bool(np.all(statuses == 1)), "iter_mean": float(iters.mean()), "iter_max": int(iters.max()),
                            "per_qp_mean": {k: float(np.mean([getattr(s, k) for s in stats])) for k in keys},
                            "phase_ms_per_qp": phase},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(probs, settings_kw)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
