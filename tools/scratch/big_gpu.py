"""One large QP on the GPU (BASELINE.json config 5 shape): python tools/scratch/big_gpu.py n m [nonconvex] -> timings, KKT residuals"""
import sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, "/root/repo")
from qpalm_amd.problems import QP, random_qp
from qpalm_amd.solver import Context, QpalmBatch
n, m = int(sys.argv[1]), int(sys.argv[2])
nonconvex = len(sys.argv) > 3 and sys.argv[3] == "nonconvex"
p = random_qp(n, m, seed=5, density_A=10.0 / n, density_M=5.0 / n)
if nonconvex:
    rng = np.random.default_rng(5)
    Q = p.Q_full().tolil()
    d = Q.diagonal()
    flip = rng.random(n) < 0.2
    Q.setdiag(np.where(flip, -0.5 * d, d))
    Ql = sp.tril(Q.tocsc()).tocsc(); Ql.sort_indices()
    p = QP(n, m, Ql.indptr.astype(np.int64), Ql.indices.astype(np.int64), Ql.data.copy(), p.Ap, p.Ai, p.Ax, p.q, p.bmin, p.bmax)
ctx = Context(0)
t0 = time.time()
bt = QpalmBatch(ctx, [p], ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0, nonconvex=1 if nonconvex else 0))
t1 = time.time()
bt.solve()
t2 = time.time()
info, s = bt.info(0), bt.stats(0)
x, y = bt.solution_of(0)
A, Qf = p.A_mat(), p.Q_full()
ax = A @ x
prim = np.max(np.maximum(p.bmin - ax, 0) + np.maximum(ax - p.bmax, 0)) / max(1.0, np.max(np.abs(ax)))
g = Qf @ x + p.q + A.T @ y
dual = np.max(np.abs(g)) / max(1.0, np.max(np.abs(Qf @ x)), np.max(np.abs(A.T @ y)))
print("n", n, "m", m, "nonconvex", nonconvex, "status", info.status.decode(), "iter", int(info.iter), "setup %.2fs solve %.2fs (kernel %.1f ms)" % (t1 - t0, t2 - t1, bt.last_solve_ms()),
      "refactor", int(s.n_refactor), "rank1", int(s.n_rank1), "sweeps", int(s.n_sweeps), "lambda", s.lobpcg_lambda, "lobpcg_iter", int(s.lobpcg_iter), "gamma", s.gamma,
      "prim %.2e dual %.2e" % (prim, dual), "ms factor %.0f update %.0f solve %.0f ls %.0f" % (s.ms_factor, s.ms_update, s.ms_solve, s.ms_linesearch))
