import sys, numpy as np
sys.path.insert(0, '.')
import torch; torch.cuda.init()
from qpalm_amd.solver import Context, QpalmBatch
from qpalm_amd.problems import random_qp
ctx = Context(0)
for n, m, slots in ((100, 200, 2), (300, 600, 2)):
    ctx.set_option("max_slots", slots)
    probs = [random_qp(n, m, seed=520 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(7)]
    bt = QpalmBatch(ctx, probs, ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
    bt.solve()
    print(n, "order after solve 1", [int(bt.ivec("queue_order", k, 1)[0]) for k in range(7)], "cost", [round(float(bt.stats(k).ms_total), 3) for k in range(7)], ctx.L.qpg_last_error())
    bt.warm_start(None, None); bt.solve()
    print(n, "order after solve 2", [int(bt.ivec("queue_order", k, 1)[0]) for k in range(7)], [int(bt.info(k).status_val) for k in range(7)])
    bt.close()
