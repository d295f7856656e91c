#!/usr/bin/env python3
"""where one large sparse QP's time goes (phase timers of k_solve): sparse_phases.py kind n [kind n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpalm_amd.problems import sparse_qp  # noqa: E402
from qpalm_amd.solver import Context, QpalmBatch  # noqa: E402

ctx = Context(0)
args = sys.argv[1:] or ["blocks", "100000", "banded", "100000"]
for kind, n in zip(args[0::2], args[1::2]):
    p = sparse_qp(int(n), kind, seed=21)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
    t0 = time.perf_counter()
    bt.solve()
    dt = time.perf_counter() - t0
    s = bt.stats(0)
    names = ("ms_factor", "ms_update", "ms_solve", "ms_linesearch")
    print(kind, n, "%.2f s, %d iterations, kernel %.0f ms:" % (dt, int(bt.info(0).iter), float(s.ms_total)), {k: round(float(getattr(s, k)), 1) for k in names},
          "dbg", [round(float(v), 1) for v in s.ms_dbg])
    bt.close()
