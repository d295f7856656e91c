#!/usr/bin/env python3
"""Where qpalm_setup's time goes for the benchmark batch (run on the GPU box): the Python side of QpalmBatch (pointer arrays),
qpg_batch_set_problems by host-thread count, qpg_batch_setup (packing + DMA + Ruiz scaling on the device)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
torch.cuda.init()
from qpalm_amd.capi import f64, i64  # noqa: E402
from qpalm_amd.problems import random_qp  # noqa: E402
from qpalm_amd.solver import Context  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = Context(0)
base = [random_qp(1000, 2000, seed=1000 + k, density_A=0.01, density_M=0.005) for k in range(64)]
probs = [base[k % 64] for k in range(B)]
L = ctx.L
t0 = time.perf_counter()
cols = [(C.c_void_p * B)() for _ in range(9)]
keep = []
for b, p in enumerate(probs):
    arrs = (i64(p.Qp), i64(p.Qi), f64(p.Qx), i64(p.Ap), i64(p.Ai), f64(p.Ax), f64(p.q), f64(p.bmin), f64(p.bmax))
    keep.append(arrs)
    for k, a in enumerate(arrs):
        cols[k][b] = a.ctypes.data
print("python side (pointer arrays) %.3f s" % (time.perf_counter() - t0))
for th in ("1", "8", "16", "32", "64", "default"):
    if th == "default":
        os.environ.pop("QPALM_HOST_THREADS", None)
    else:
        os.environ["QPALM_HOST_THREADS"] = th
    h = C.c_void_p()
    st = ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    t0 = time.perf_counter()
    assert L.qpg_batch_create(ctx.h, B, 1000, 2000, 20000, 6000, C.byref(st), C.byref(h)) == 0   # (starts the device arena allocation on a thread)
    t1 = time.perf_counter()
    rc = L.qpg_batch_set_problems(h, 0, B, None, None, *cols[:7], None, cols[7], cols[8])
    t2 = time.perf_counter()
    rc2 = L.qpg_batch_setup(h)
    t3 = time.perf_counter()
    print("host threads %7s: create %.3f s, set_problems %.3f s (rc %d), batch_setup %.3f s (rc %d), total %.3f s" % (th, t1 - t0, t2 - t1, rc, t3 - t2, rc2, t3 - t0))
    t4 = time.perf_counter()
    L.qpg_batch_destroy(h)
    print("                       destroy %.3f s" % (time.perf_counter() - t4))
