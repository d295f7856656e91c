import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
from qpalm_amd.problems import random_qp
from qpalm_amd.solver import Context, QpalmBatch
n = int(sys.argv[1]); B = int(sys.argv[2]); m = 2 * n
ctx = Context(0, lib_path=os.environ.get("DBG_LIB"))
probs = [random_qp(n, m, seed=1000 + k, density_A=0.01 if n >= 400 else max(0.01, 4.0 / n), density_M=0.005 if n >= 400 else max(0.005, 2.0 / n)) for k in range(B)]
bt = QpalmBatch(ctx, probs, ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
bt.warm_start(None, None)
for it in range(40):
    bt.iterate(1)
    st = [(int(bt.stats(b).last_kind), int(bt.stats(b).last_fact)) for b in range(min(B, 4))]
    print("it", it, st, bt.statuses()[:4], flush=True)
    if bt.num_unfinished() == 0: break
print("done")
