#!/usr/bin/env python3
"""phase split (mean ms per QP inside the kernel) of a batch of sparse QPs: sparse_batch_phases.py [kind n B]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpalm_amd.problems import sparse_qp  # noqa: E402
from qpalm_amd.solver import Context, QpalmBatch  # noqa: E402

kind, n, B = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("banded", 2000, 2048)
ctx = Context(0)
ctx.set_option("sparse_factor", 1)
distinct = [sparse_qp(n, kind, seed=700 + k) for k in range(32)]
bt = QpalmBatch(ctx, [distinct[k % 32] for k in range(B)], ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
bt.solve()
bt.warm_start(None, None)
t0 = time.perf_counter()
bt.solve()
dt = time.perf_counter() - t0
st = bt.stats_all()
mean = lambda f: float(np.mean([f(s) for s in st]))
print(kind, n, B, "%.3f s = %.0f QP/s; iterations %.1f, refactor %.1f, rank1 %.1f; ms per QP: total %.1f factor %.1f update %.1f solve %.1f linesearch %.1f (sort %.1f) residuals %.1f" % (
    dt, B / dt, float(np.mean([int(i.iter) for i in bt.infos()])), mean(lambda s: s.n_refactor), mean(lambda s: s.n_rank1), mean(lambda s: s.ms_total), mean(lambda s: s.ms_factor),
    mean(lambda s: s.ms_update), mean(lambda s: s.ms_solve), mean(lambda s: s.ms_linesearch), mean(lambda s: s.ms_dbg[15]), mean(lambda s: s.ms_dbg[12])))
