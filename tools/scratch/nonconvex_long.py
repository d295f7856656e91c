#!/usr/bin/env python3
"""fuzz case 682 / 1 (nonconvex, n = 224): where does the engine's trajectory end when it is given more iterations?"""
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qpalm_amd.solver import Context, QpalmBatch  # noqa: E402
from tests.fuzz_cases import cases  # noqa: E402
import oracle.binding as ob  # noqa: E402

ctx = Context(0)
for it, p, st, warm, meta in cases(682, 2, 130, 600, dict(factorization_method=1, nonconvex=1, q_shift=1.0)):
    if it != 1:
        continue
    for mi in (10000, 100000):
        st2 = dict(st, max_iter=mi)
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**st2))
        if warm is not None:
            bt.warm_start(warm[0][None, :], warm[1][None, :])
        bt.solve()
        info = bt.info(0)
        x, y = bt.solution()
        A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n)); Ql = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(p.n, p.n)); Q = Ql + sp.tril(Ql, -1).T
        ax = A @ x[0]
        print("engine max_iter %d: status %d iter %d outer %d objective %.6e pri %.2e dua %.2e | feasibility %.2e stationarity %.2e gamma %.3e" % (
            mi, int(info.status_val), int(info.iter), int(info.iter_out), info.objective, info.pri_res_norm, info.dua_res_norm,
            np.max(np.maximum(p.bmin - ax, 0) + np.maximum(ax - p.bmax, 0)), np.max(np.abs(Q @ x[0] + p.q + A.T @ y[0])), float(bt.stats(0).gamma)))
        bt.close()
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    if warm is not None:
        o.warm_start(warm[0], warm[1])
    o.solve()
    print("oracle: status %d iter %d outer %d objective %.6e pri %.2e dua %.2e" % (o.status_val, int(o.info.iter), int(o.info.iter_out), o.info.objective, o.info.pri_res_norm, o.info.dua_res_norm))
