#!/usr/bin/env python3
"""fuzz LP case 701 / 114: the first iteration at which the engine's iterate stops being finite, and the state around it"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qpalm_amd.solver import Context, QpalmBatch  # noqa: E402
from tests.fuzz_cases import cases  # noqa: E402

ctx = Context(0)
if len(sys.argv) > 1:
    ctx.set_option("sequential_rank_sums", int(sys.argv[1]))
for it, p, st, warm, meta in cases(701, 115, 70, 400, dict(lp=1, factorization_method=1)):
    if it != 114:
        continue
    def fresh():
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
        if warm is not None:
            bt.warm_start(warm[0][None, :], warm[1][None, :])
        bt.begin_solve()
        return bt
    bt = fresh()
    k = 0
    while k < 10000:
        bt.iterate(50); k += 50
        if not np.all(np.isfinite(bt.vec("x", 0)[:p.n])) or int(bt.info(0).status_val) != -10:
            break
    print("after", k, "iterations: status", int(bt.info(0).status_val), "finite", bool(np.all(np.isfinite(bt.vec("x", 0)[:p.n]))))
    bt.close()
    if k >= 10000:
        break
    bt = fresh()
    bt.iterate(k - 50)
    for j in range(k - 50, k):
        bt.iterate(1)
        s = bt.stats(0)
        x = bt.vec("x", 0)[:p.n]; d = bt.vec("d", 0)[:p.n]
        L, D = bt.factor(0)
        fin = np.all(np.isfinite(x))
        print(j, "kind", int(s.last_kind), "fact", int(s.last_fact), "enter/leave %d/%d" % (int(s.nb_enter), int(s.nb_leave)), "tau %.3e gamma %.3e" % (float(s.tau), float(s.gamma)),
              "min|D| %.3e min D %.3e nonfinite D %d L %d d %d x %d" % (np.nanmin(np.abs(D)), np.nanmin(D), int(np.sum(~np.isfinite(D))), int(np.sum(~np.isfinite(L))), int(np.sum(~np.isfinite(d))), int(np.sum(~np.isfinite(x)))))
        if not fin:
            break
