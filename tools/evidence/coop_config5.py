#!/usr/bin/env python3
"""BASELINE.json config 5 as written (nonconvex random QP, n = 5000, LOBPCG front-end) on one MI355X in coop mode, with the
host-side phase profile of the solve (QPALM_COOP_PROFILE: synchronises after every phase, a few per cent slower).
usage: coop_config5.py [n [coop_rank_threshold ...]]   (default 5000 with the default policy -2)"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

torch.cuda.init()
os.environ.setdefault("QPALM_COOP_PROFILE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpalm_amd.problems import random_qp  # noqa: E402
from qpalm_amd.solver import Context, QpalmBatch  # noqa: E402

ctx = Context(0, lib_path=os.environ.get("QPALM_LIB") or None)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = n
p = random_qp(n, m, seed=55, density_A=10.0 / n, density_M=5.0 / n)
Q = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(n, n)).tolil()
for j in range(0, n, 5):
    Q[j, j] = Q[j, j] - 2.5 * abs(Q[j, j])
Q = sp.csc_matrix(Q)
Q.sort_indices()
p2 = type(p)(n, m, Q.indptr.astype(np.int64), Q.indices.astype(np.int64), Q.data.copy(), p.Ap, p.Ai, p.Ax, p.q, p.bmin, p.bmax)
st = dict(eps_abs=1e-5, eps_rel=1e-5, verbose=0, nonconvex=1, max_iter=20000)
for thr in [int(a) for a in sys.argv[2:]] or [-2]:
    ctx.set_option("coop", 1)
    ctx.set_option("coop_graphs", int(os.environ.get("QPALM_COOP_GRAPHS", "1")))
    ctx.set_option("coop_rank_threshold", thr)
    bt = QpalmBatch(ctx, [p2], ctx.default_settings(**st))
    t0 = time.perf_counter()
    bt.solve()
    dt = time.perf_counter() - t0
    s, info = bt.stats(0), bt.info(0)
    print("n=%d m=%d nonconvex, coop_rank_threshold=%d: %.2f s, %d iterations (%d outer), status %d, refactor %d, rank-1 %d, LOBPCG %d iterations, lambda %.6f" % (
        n, m, thr, dt, info.iter, info.iter_out, info.status_val, s.n_refactor, s.n_rank1, s.lobpcg_iter, s.lobpcg_lambda))
    sys.stdout.flush()
    bt.close()
