"""GPU-only tests at BASELINE.json's full size (n = 1000, m = 2000: the k_solve<2> instantiation that
bench.py measures) and above it (n = 1100: k_solve<4>).

 * a few QPs of the benchmark batch against the CPU oracle (same tolerances as tests/test_parity.py,
   iteration / refactor / rank-1 counts and active sets exact);
 * size-independent properties on a larger batch, checked with plain numpy on the host:
   KKT conditions of the returned (x, y), determinism (same QP at different batch positions and in a
   queued run gives bit-identical results), idempotence of a warm-started re-solve.
"""
import numpy as np
import pytest

from oracle import binding as ob
from qpalm_amd.problems import random_qp
from qpalm_amd.solver import Context, QpalmBatch

pytestmark = pytest.mark.gpu
RTOL = 1e-9
ST = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1.0, np.max(np.abs(b)))


@pytest.fixture(scope="module")
def hip():
    c = Context(0)
    assert c.backend == "gfx950-hip"
    return c


def bench_qp(n, m, seed):
    return random_qp(n, m, seed=seed, density_A=0.01, density_M=0.005)  # bench.py's generator settings


def dense(p):
    import scipy.sparse as sp
    n, m = p.n, p.m
    A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(m, n)).toarray()
    Ql = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(n, n)).toarray()
    Q = Ql + Ql.T - np.diag(np.diag(Ql))
    return Q, A


@pytest.mark.parametrize("n,m,nb", [(1000, 2000, 32), (1100, 2200, 8)])
def test_full_size_matches_oracle(hip, n, m, nb):
    """32 QPs of the benchmark batch (k_solve<2>, 16 ranks per sweep) and 8 at n = 1100 (k_solve<4>, 8 ranks per sweep)
    against the oracle; the oracle solves run on a thread pool (ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    probs = [bench_qp(n, m, 1000 + k) for k in range(nb)]
    bt = QpalmBatch(hip, probs, hip.default_settings(**ST))
    bt.solve()
    xs, ys = bt.solution()

    def run_oracle(p):
        o = ob.OracleQP(*p.args(), c=p.c, settings=ob.default_settings(**ST))
        o.solve()
        return o

    with ThreadPoolExecutor(8) as ex:
        oracles = list(ex.map(run_oracle, probs))
    K = 16 if n <= 1024 else 8
    for k, (p, o) in enumerate(zip(probs, oracles)):
        info, s = bt.info(k), bt.stats(k)
        assert int(info.status_val) == o.status_val == 1
        assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
        assert int(s.n_refactor) == o.counter("n_refactor") and int(s.n_rank1) == o.counter("n_rank1")
        assert rel(xs[k], o.x) <= RTOL and rel(ys[k], o.y) <= RTOL
        assert abs(info.pri_res_norm - o.info.pri_res_norm) <= 1e-8 and abs(info.dua_res_norm - o.info.dua_res_norm) <= 1e-8
        assert np.array_equal(bt.ivec("active", k), o.ivec("active"))
        # work counters written by the sweep itself: at least ceil(ranks / K) sweeps, at most one more per update call
        assert int(s.n_sweeps) * K >= int(s.n_rank1) and int(s.n_sweeps) >= 1
        assert 0 < int(s.sweep_entries) <= int(s.n_sweeps) * (n * (n + 1) // 2)
        o.cleanup()


def test_full_size_properties(hip):
    n, m, nb = 1000, 2000, 48
    probs = [bench_qp(n, m, 5000 + k) for k in range(nb)]
    probs[17] = probs[3]                     # the same QP twice in the batch
    bt = QpalmBatch(hip, probs, hip.default_settings(**ST))
    bt.solve()
    xs, ys = bt.solution()
    assert np.all(bt.statuses() == 1)
    # KKT conditions on the host, unscaled problem data
    for k in (0, 3, 11, 29, 47):
        p = probs[k]
        Q, A = dense(p)
        x, y = xs[k], ys[k]
        ax = A @ x
        scale_p = max(1.0, np.max(np.abs(ax)))
        assert np.max(np.maximum(p.bmin - ax, 0) + np.maximum(ax - p.bmax, 0)) <= 1e-5 * scale_p       # primal feasibility
        grad = Q @ x + p.q + A.T @ y
        scale_d = max(1.0, np.max(np.abs(Q @ x)), np.max(np.abs(p.q)), np.max(np.abs(A.T @ y)))
        assert np.max(np.abs(grad)) <= 1e-5 * scale_d                                                   # stationarity
        slack_lo, slack_hi = ax - p.bmin, p.bmax - ax
        assert np.all((y <= 1e-7) | (slack_hi <= 1e-4 * scale_p)) and np.all((y >= -1e-7) | (slack_lo <= 1e-4 * scale_p))  # complementarity
    # determinism: same QP, different batch position
    assert np.array_equal(xs[17], xs[3]) and np.array_equal(ys[17], ys[3])
    assert int(bt.info(17).iter) == int(bt.info(3).iter)
    # ... and through the work queue (4 slots for 48 QPs)
    hip.set_option("max_slots", 4)
    bq = QpalmBatch(hip, probs, hip.default_settings(**ST))
    hip.set_option("max_slots", 512)
    bq.solve()
    xq, yq = bq.solution()
    assert np.array_equal(xq, xs) and np.array_equal(yq, ys)
    # idempotence: warm start at the solution -> the same point again, in fewer iterations than the cold start
    # (penalties restart from sigma_init, so it is not a single iteration)
    it0 = [int(bt.info(k).iter) for k in range(nb)]
    bt.warm_start(xs, ys)
    bt.solve()
    x2, y2 = bt.solution()
    assert np.all(bt.statuses() == 1)
    assert all(int(bt.info(k).iter) < it0[k] for k in range(nb))
    # two eps = 1e-6 solutions of the same QP: close, not identical
    assert rel(x2, xs) <= 1e-4 and rel(y2, ys) <= 1e-3, (rel(x2, xs), rel(y2, ys))


def test_device_views_for_the_rccl_gather(hip):
    """bench.py gathers solution_x / solution_y with torch.distributed straight from HBM: the zero-copy
    torch views must alias the batch's device arrays."""
    import torch
    from qpalm_amd.dist import device_view
    n, m, nb = 1000, 2000, 4
    probs = [bench_qp(n, m, 7000 + k) for k in range(nb)]
    bt = QpalmBatch(hip, probs, hip.default_settings(**ST))
    bt.solve()
    xs, ys = bt.solution()
    tx = device_view(bt, "solution_x", (nb, n), "cuda:0")
    ty = device_view(bt, "solution_y", (nb, m), "cuda:0")
    assert tx.dtype == torch.float64 and tx.is_cuda and tuple(tx.shape) == (nb, n)
    assert np.array_equal(tx.cpu().numpy(), xs) and np.array_equal(ty.cpu().numpy(), ys)
    # what rank 0 does with the gathered buffers
    g = [torch.empty_like(tx) for _ in range(2)]
    g[0].copy_(tx); g[1].copy_(tx)
    torch.cuda.synchronize()
    assert np.array_equal(g[1].cpu().numpy(), xs)
