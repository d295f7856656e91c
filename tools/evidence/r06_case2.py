import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch; torch.cuda.init()
import numpy as np
from qpalm_amd.solver import Context, QpalmBatch
from tests.fuzz_cases import cases
ctx = Context(0)
for it, p, st, warm, meta in cases(721, 8, 70, 300, dict(q_scale=1e-10, factorization_method=1)):
    if it != 7:
        continue
    for sw in (1, 0):
        for mode in (-1, 1, 0):
            ctx.set_option("small_workgroups", sw)
            ctx.set_option("sequential_rank_sums", mode)
            bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
            if warm is not None:
                bt.warm_start(warm[0][None, :], warm[1][None, :])
            bt.solve()
            s = bt.stats(0)
            print("small_workgroups", sw, "mode", mode, "threads", bt.launch_shape()[1], "status", int(bt.info(0).status_val), "iter", int(bt.info(0).iter), "obj", float(bt.info(0).objective),
                  "seq cols", int(s.n_seq_columns), "of", int(s.n_sweep_columns), "rank1", int(s.n_rank1), "refactor", int(s.n_refactor), "guard", int(s.n_guard_refactor), flush=True)
            bt.close()
