"""Nonconvex front-end: lobpcg + set_settings_nonconvex on the device at setup
(src/nonconvex.c:29-183), the nonconvex branch of the loop (src/qpalm.c:586-611,655) and the indefinite LDL'.

Golden: tests/src/test_nonconvex_qp.c:117-126 (gamma within 10 % of 1/0.0021544347, eigenvalue under-approximated), in the
three factorization modes the reference runs (:132-139).  The dot products of LOBPCG are summed in the reference's order on
the device, so lambda, the LOBPCG iteration count and gamma are compared with the oracle to the last bits (<= 1e-12) and
the iterates of the solve to the usual 1e-9.  Factors of up to 8192 rows are supported; BASELINE.json config 5 itself (n = 5000) is tests/test_coop.py::test_config5_nonconvex_n5000."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import binding as ob
from qpalm_amd.problems import QP, random_qp
from qpalm_amd.solver import QpalmBatch
from tests.helpers import STATUS
from tests.test_parity import RTOL, rel, sizes


from qpalm_amd.problems import fixture_qp


def indefinite_qp(n, m, seed):
    pr = random_qp(n, m, seed=seed, density_A=max(0.03, 4.0 / n), density_M=max(0.02, 2.0 / n))
    rng = np.random.default_rng(seed)
    Q = pr.Q_full().toarray()
    Q = Q - 1.5 * np.diag(np.diag(Q)) * (rng.random(n) < 0.3)      # some negative diagonal entries: indefinite
    Ql = sp.tril(sp.csc_matrix(Q)).tocsc()
    Ql.sort_indices()
    return QP(n, m, Ql.indptr.astype(np.int64), Ql.indices.astype(np.int64), Ql.data.copy(), pr.Ap, pr.Ai, pr.Ax, pr.q, pr.bmin, pr.bmax), Q


def test_restated_rand_is_the_c_librarys():
    out = (C.c_int * 400)()
    libc = C.CDLL("libc.so.6")
    for seed in (1, 12345):
        ob.lib().oq_rand_sequence(seed, 400, out)
        libc.srand(seed)
        assert list(out) == [libc.rand() for _ in range(400)]


@pytest.mark.parametrize("method", [2, 0, 1])   # KKT_OR_SCHUR, KKT, SCHUR as in the reference's suite
def test_reference_nonconvex_qp(ctx, golden, method):
    p = fixture_qp(golden["problems"]["nonconvex_qp"])   # tests/src/test_nonconvex_qp.c:40-95
    st = dict(eps_abs=1e-6, eps_rel=1e-6, nonconvex=1, scaling=0, max_rank_update_fraction=1.0, factorization_method=method, verbose=0)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    bt.solve()
    s, info = bt.stats(0), bt.info(0)
    assert int(info.status_val) == STATUS["SOLVED"]
    e = golden["expect"]["nonconvex_qp"]
    lam_min = -e["lambda_min"]
    assert abs(s.gamma - 1.0 / lam_min) <= e["gamma_rel_tol"] / lam_min        # inverse of the lowest eigenvalue
    assert 1 / s.gamma > lam_min                                  # the eigenvalue is under-approximated
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    assert abs(s.lobpcg_lambda - o.scalar("lobpcg_lambda")) <= 1e-12 * abs(o.scalar("lobpcg_lambda"))
    assert int(s.lobpcg_iter) == o.counter("n_lobpcg_iter") and int(s.nonconvex) == o.counter("nonconvex") == 1
    o.solve()
    assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
    assert rel(bt.solution()[0][0], o.x) <= RTOL and rel(bt.solution()[1][0], o.y) <= RTOL


def test_indefinite_qps_in_a_batch(ctx):
    """indefinite and convex members in one nonconvex batch: per-QP gamma = 1/|lambda| where lambda < 0, the convex member
    falls back to settings->nonconvex = FALSE (nonconvex.c:179-182); scaled and unscaled"""
    n, m = sizes(ctx, (30, 50), (160, 320))
    probs, mats = [], []
    for k in range(sizes(ctx, 1, 4)):
        p, Q = indefinite_qp(n, m, 50 + k)
        probs.append(p); mats.append(Q)
    probs.append(random_qp(n, m, seed=77, density_A=max(0.03, 4.0 / n), density_M=max(0.02, 2.0 / n)))   # convex
    for scaling in (0, 10):
        st = dict(eps_abs=1e-6, eps_rel=1e-6, nonconvex=1, scaling=scaling, verbose=0)
        bt = QpalmBatch(ctx, probs, ctx.default_settings(**st))
        bt.solve()
        xs, ys = bt.solution()
        for k, p in enumerate(probs):
            o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
            s = bt.stats(k)
            assert abs(s.lobpcg_lambda - o.scalar("lobpcg_lambda")) <= 1e-11 * max(1.0, abs(o.scalar("lobpcg_lambda"))), (k, s.lobpcg_lambda)
            assert int(s.lobpcg_iter) == o.counter("n_lobpcg_iter") and int(s.nonconvex) == o.counter("nonconvex")
            if k < len(mats) and scaling == 0:
                ev = np.linalg.eigvalsh(mats[k])[0]
                assert ev < 0 and s.lobpcg_lambda <= ev + 1e-6 and abs(s.lobpcg_lambda - ev) <= 1e-3 * abs(ev)   # a lower bound close to lambda_min
            assert int(s.nonconvex) == (1 if k < len(mats) else 0)
            o.solve()
            info = bt.info(k)
            assert int(info.status_val) == o.status_val
            assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out), (k, info.iter, o.info.iter)
            assert rel(xs[k], o.x) <= 1e-8 and rel(ys[k], o.y) <= 1e-8
            if int(s.nonconvex):
                assert abs(s.gamma - 1.0 / abs(s.lobpcg_lambda)) <= 1e-12 * s.gamma
