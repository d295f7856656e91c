import sys, numpy as np
sys.path.insert(0, "/root/repo")
from qpalm_amd.problems import random_qp
from qpalm_amd.solver import Context, QpalmBatch
op = sys.argv[1]; n = int(sys.argv[2]); m = 2 * n
import os
ctx = Context(0, lib_path=os.environ.get("DBG_LIB"))
probs = [random_qp(n, m, seed=5 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(2)]
bt = QpalmBatch(ctx, probs, ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
bt.warm_start(None, None)
if op == "solve": bt.solve()
elif op == "iter1": bt.iterate(1)
elif op == "iters":
    for it in range(40):
        bt.iterate(1); bt.L.qpg_batch_sync(bt.h) if hasattr(bt.L, "qpg_batch_sync") else None
        print("it", it, [int(bt.info(b).iter) for b in range(2)], [int(bt.stats(b).n_rank1) for b in range(2)], [(int(bt.stats(b).n_refactor), int(bt.stats(b).n_sigma_updates), int(bt.stats(b).n_boost_gamma), int(bt.stats(b).last_kind), int(bt.stats(b).last_fact), int(bt.stats(b).nb_active)) for b in range(2)], bt.statuses(), flush=True)
elif op == "ops":
    for it in range(11): bt.iterate(1)
    print("at it 11", flush=True)
    for o in ("compute_residuals", "set_active_constraints", "ldlsolveLD_neg_dphi"):
        bt.op(o, 0); print(o, "ok", flush=True)
    print("ls", bt.exact_linesearch(0), flush=True)
    bt.iterate(1); print("iterate ok", flush=True)
elif op == "ls": print(bt.exact_linesearch(0))
elif op == "ldlall": print(bt.ldlsolve_all(1))
else: bt.op(op, 0)
bt.sync() if hasattr(bt, "sync") else None
print(op, n, "ok", bt.statuses())
