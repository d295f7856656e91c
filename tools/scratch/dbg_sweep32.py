import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.cuda.init()
from qpalm_amd.problems import random_qp
from qpalm_amd.solver import Context, QpalmBatch
lib = sys.argv[1] if len(sys.argv) > 1 else None
ctx = Context(0, lib_path=lib)
n, m = 500, 1300
p = random_qp(n, m, seed=4321, density_A=0.01, density_M=0.005)
act = (np.random.default_rng(99).random(m) < 0.3).astype(np.int64)
res = {}
for sr in (16, 32):
    ctx.set_option("sweep_ranks", sr)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
    bt.iterate(3); bt.set_ivec("active", act); bt.op("ldlcholQAtsigmaA")
    enter = np.where(act == 0)[0][:20]
    bt.set_ivec("enter", enter); bt.set_scalar("nb_enter", len(enter)); bt.set_scalar("nb_leave", 0)
    bt.op("ldlupdate_entering_constraints")
    res[sr] = bt.factor()
L32, D32 = res[32]; L16, D16 = res[16]
d = np.abs(L32 - L16)
print("lib", lib, "max |L32-L16| %.3e  max |D32-D16| %.3e  nan %d" % (np.nanmax(d), np.nanmax(np.abs(D32 - D16)), int(np.isnan(L32).sum())))
bad = np.argwhere(d > 1e-12)
if len(bad): print("first bad entries (row, col):", bad[:6].tolist(), "count", len(bad), "cols with errors:", sorted(set(bad[:, 1].tolist()))[:20])
