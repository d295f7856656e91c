"""Bisection of the sparse factor's LDS forms on the hardware: one library build and one mode per process (a fault ends only that process)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from qpalm_amd.solver import Context, QpalmBatch
from qpalm_amd.problems import sparse_qp

lib, kind, n, ordering, lds = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
ctx = Context(0, lib_path=None if lib == "-" else lib)
p = sparse_qp(n, kind, seed=11)
ctx.set_option("sparse_factor", 1); ctx.set_option("sparse_ordering", ordering)
try:
    ctx.set_option("sparse_lds", lds)
except Exception as e:
    print("(no sparse_lds option in this build)")
bt = QpalmBatch(ctx, [p], ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
import time
t0 = time.time()
bt.solve()
x, y = bt.solution()
print(lib, kind, n, ordering, "lds", lds, "status", int(bt.info(0).status_val), "iter", int(bt.info(0).iter), "x", float(np.sum(x[0])), "s", round(time.time() - t0, 3), flush=True)
bt.close()
