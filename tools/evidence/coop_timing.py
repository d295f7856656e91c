#!/usr/bin/env python3
"""One large QP alone on the chip: coop mode against the one-workgroup engine (BASELINE.json configs 2 and 5 as written).
usage: coop_timing.py [n ...]   (default 1000 2500 5000)"""
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from qpalm_amd.problems import random_qp  # noqa: E402
from qpalm_amd.solver import Context, QpalmBatch  # noqa: E402

ctx = Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [1000, 2500, 5000]
for n in sizes:
    m = 2 * n if n <= 2500 else n
    p = random_qp(n, m, seed=1000, density_A=min(0.01, 10.0 / n), density_M=min(0.005, 5.0 / n))
    res = {}
    for coop in ((1, 0) if n <= 2500 else (1,)):
        ctx.set_option("coop", coop)
        bt = QpalmBatch(ctx, [p], ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
        bt.solve()
        t = []
        for _ in range(2):
            bt.warm_start(None, None)
            t0 = time.perf_counter()
            bt.solve()
            t.append(time.perf_counter() - t0)
        x, y = bt.solution()
        s = bt.stats(0)
        res[coop] = (min(t), int(bt.info(0).iter), int(bt.info(0).status_val), x[0].copy(), int(s.n_refactor), int(s.n_factor_Q), int(s.n_rank1))
        print("n=%d m=%d coop=%d: %.1f ms per solve, %d iterations, status %d, refactor %d + %d, rank-1 %d" % (
            n, m, coop, 1e3 * min(t), res[coop][1], res[coop][2], res[coop][4], res[coop][5], res[coop][6]))
        sys.stdout.flush()
        bt.close()
    if 0 in res:
        print("   coop vs one workgroup: max |dx| %.2e, speed-up %.1fx" % (np.max(np.abs(res[1][3] - res[0][3])), res[0][0] / res[1][0]))
