"""The KKT path with row add / delete (SURVEY section 8 rows a17 = f1): FACTORIZE_KKT through the same C ABI.

Reference: src/newton.c:22-95 (policy + iterative refinement), src/solver_interface.c:119-247 (KKT assembly, ladel_row_add /
ladel_row_del, kkt_solve), chooser :20-75.  LADEL is absent from the reference tree, so the oracle restates the published
algorithms (oracle/qpalm_oracle.c, "KKT path"); the reference's own tests pin this path only end to end (they run under the
LADEL build, tests/src/test_basic_qp.c etc.), which is what the golden comparisons below do.

Tolerances: iterations, refactorisation / row-operation counts, statuses and active sets exact; x, y <= 1e-9 relative."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import binding as ob
from qpalm_amd.problems import fixture_qp, random_mpc_qp, random_qp
from qpalm_amd.solver import QpalmBatch
from tests.helpers import STATUS
from tests.test_parity import RTOL, gsettings, rel, sizes

KKT = dict(factorization_method=0)


def _pair(ctx, p, st):
    st = dict(st, verbose=0, **KKT)
    o = ob.OracleQP(*p.args(), c=p.c, settings=ob.default_settings(**st))
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    return o, bt


def _check(o, bt, k=0):
    info, s = bt.info(k), bt.stats(k)
    assert o.counter("kkt_mode") == 1
    assert int(info.status_val) == o.status_val
    assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
    assert int(s.n_refactor) == o.counter("n_refactor")
    assert int(s.n_rank1) == o.counter("n_row_add") + o.counter("n_row_del")
    x, y = bt.solution()
    assert rel(x[k], o.x) <= RTOL and rel(y[k], o.y) <= RTOL
    assert np.array_equal(bt.ivec("active", k), o.ivec("active"))


@pytest.mark.parametrize("name,over", [
    ("basic_qp", dict()), ("basic_qp", dict(scaling=0)), ("basic_qp", dict(proximal=0, scaling=2)),
    ("basic_qp", dict(proximal=0, scaling=0)), ("basic_qp", dict(sigma_max=1e3)),
    ("medium_qp", dict()), ("degen_hess", dict()), ("ls_qp", dict()),
])
def test_reference_solutions_in_kkt_mode(ctx, golden, name, over):
    """the golden solutions of the reference's suites with factorization_method = FACTORIZE_KKT"""
    e = golden["expect"][name]
    p = fixture_qp(golden["problems"][name])
    o, bt = _pair(ctx, p, gsettings(ctx, golden, name, **over))
    o.solve(); bt.solve()
    assert int(bt.info(0).status_val) == STATUS["SOLVED"]
    x = bt.solution()[0][0]
    if "rel_tol" in e:
        for a, b in zip(x, e["solution"]):
            assert abs(a - b) <= abs(e["rel_tol"] * b)
    else:
        assert np.max(np.abs(x - e["solution"])) <= e["abs_tol"]
    _check(o, bt)


@pytest.mark.parametrize("name,status", [("prim_inf_qp", "PRIMAL_INFEASIBLE"), ("dua_inf_qp", "DUAL_INFEASIBLE")])
def test_reference_infeasible_in_kkt_mode(ctx, golden, name, status):
    for k in range(4):
        st = gsettings(ctx, golden, name, **golden["expect"][name]["variants"][k])
        o, bt = _pair(ctx, fixture_qp(golden["problems"][name]), st)
        bt.solve()
        assert int(bt.info(0).status_val) == STATUS[status]


def test_random_and_mpc_qps_in_kkt_mode(ctx):
    n, m = sizes(ctx, (40, 80), (150, 300))
    st = dict(eps_abs=1e-6, eps_rel=1e-6)
    probs = [random_qp(n, m, seed=900 + k, density_A=max(0.02, 4.0 / n), density_M=max(0.01, 2.0 / n)) for k in range(sizes(ctx, 2, 4))]
    probs.append(random_mpc_qp(T=sizes(ctx, 3, 10), nx=sizes(ctx, 4, 10), nu=sizes(ctx, 2, 5), seed=7))
    adds = 0
    for p in probs:
        o, bt = _pair(ctx, p, st)
        o.solve(); bt.solve()
        assert o.status_val == STATUS["SOLVED"]
        _check(o, bt)
        adds += o.counter("n_row_add") + o.counter("n_row_del")
    assert adds > 0, "no row was added or deleted: the update path did not run"


def test_kkt_batch_and_dual_termination(ctx, golden):
    """a batch (work queue: 2 slots for 4 QPs) in KKT mode, and dual termination next to it (LD_Q is a second n x n factor)"""
    n, m = sizes(ctx, (24, 48), (100, 200))
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, enable_dual_termination=1, **KKT)
    probs = [random_qp(n, m, seed=40 + k, density_A=max(0.02, 4.0 / n), density_M=max(0.01, 2.0 / n)) for k in range(4)]
    ctx.set_option("max_slots", 2)
    try:
        bt = QpalmBatch(ctx, probs, ctx.default_settings(**st))
        bt.solve()
    finally:
        ctx.set_option("max_slots", 512)
    for k, p in enumerate(probs):
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
        o.solve()
        _check(o, bt, k)
        info = bt.info(k)
        assert abs(info.dual_objective - o.info.dual_objective) <= 1e-9 * max(1.0, abs(o.info.dual_objective))
        assert abs(info.objective - info.dual_objective) <= 1e-4 * max(1.0, abs(info.objective))


def test_row_add_delete_keeps_a_valid_factor(ctx):
    """Property (no oracle): after every Newton step that changed the factor by row additions / deletions, L D L' equals
    the KKT matrix of the CURRENT active set, assembled with numpy from the device's own scaled data."""
    n, m = sizes(ctx, (30, 60), (120, 240))
    p = random_qp(n, m, seed=321, density_A=max(0.03, 4.0 / n), density_M=max(0.01, 2.0 / n))
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, max_rank_update_fraction=1.0, **KKT)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    nzA, nzQ = int(p.Ap[-1]), int(p.Qp[-1])
    checked = 0
    for it in range(60):
        bt.iterate(1)
        s = bt.stats(0)
        if int(s.last_kind) == 0 and int(s.last_fact) == 2:
            A = sp.csc_matrix((bt.named_vec("A_values", nzA), p.Ai, p.Ap), shape=(m, n)).toarray()
            Ql = sp.csc_matrix((bt.named_vec("Q_values", nzQ), p.Qi, p.Qp), shape=(n, n)).toarray()
            Q = np.tril(Ql) + np.tril(Ql, -1).T
            act = bt.ivec("active").astype(bool)      # == active_old after the step
            sig_inv = bt.vec("sigma_inv")
            K = np.eye(n + m)
            K[:n, :n] = Q + np.eye(n) / s.gamma
            for k in np.where(act)[0]:
                K[n + k, :n] = A[k]
                K[:n, n + k] = A[k]
                K[n + k, n + k] = -sig_inv[k]
            L, D = bt.factor_rows(n + m)
            R = (L * D) @ L.T
            assert np.max(np.abs(R - K)) <= 1e-9 * max(1.0, np.max(np.abs(K))), (it, np.max(np.abs(R - K)))
            checked += 1
        if bt.num_unfinished() == 0:
            break
    assert int(bt.info(0).status_val) == STATUS["SOLVED"] and checked >= 2
