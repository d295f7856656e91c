#!/usr/bin/env python3
"""coop mode, one QP alone: the rank-update sweep as one launch (coop_updates = 2) against one launch per block column (= 1), under the
reference's refactorise-or-update rule (coop_rank_threshold = -1: many updates) and the cost-based one (-2).  QPALM_COOP_PROFILE=1 for
the phase split on stderr.  usage: coop_sweep_ab.py [n ...]"""
import os
import sys
import time

import numpy as np
import torch

torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpalm_amd.problems import random_qp  # noqa: E402
from qpalm_amd.solver import Context, QpalmBatch  # noqa: E402

ctx = Context(0)
ctx.set_option("coop", 1)
sizes = [int(a) for a in sys.argv[1:]] or [1000, 2500, 5000]
for n in sizes:
    m = 2 * n if n <= 2500 else n
    p = random_qp(n, m, seed=1000, density_A=min(0.01, 10.0 / n), density_M=min(0.005, 5.0 / n))
    ref = None
    for policy in (-1, -2):
        for cu in (1, 2):
            ctx.set_option("coop_updates", cu)
            ctx.set_option("coop_rank_threshold", policy)
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
            same = ""
            if policy == -1:
                if cu == 1:
                    ref = (x[0].copy(), y[0].copy())
                else:
                    same = " bit-identical to coop_updates=1: %s" % (np.array_equal(x[0], ref[0]) and np.array_equal(y[0], ref[1]))
            print("n=%d policy %d coop_updates=%d: %.1f ms per solve, %d iterations, status %d, refactor %d + %d, rank-1 %d, sweeps %d%s" % (
                n, policy, cu, 1e3 * min(t), int(bt.info(0).iter), int(bt.info(0).status_val), int(s.n_refactor), int(s.n_factor_Q), int(s.n_rank1), int(s.n_sweeps), same))
            sys.stdout.flush()
            bt.close()
