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


def test_boundary_kkt_operations(ctx):
    """The five KKT entry points of solver_interface.h:82-126 one by one through the C ABI (qpg_kkt_form / _factorize /
    _update_entering_constraints / _update_leaving_constraints / _solve), on a state taken from the middle of a solve, each
    against the oracle's restatement of the same operation: the formed matrix against a numpy assembly, the factor entry by
    entry after the factorisation, after the row additions and after the row deletions, and sol_kkt / d after the solve."""
    import ctypes as C
    n, m = sizes(ctx, (30, 50), (300, 500))
    p = random_qp(n, m, seed=4321, density_A=max(0.02, 4.0 / n), density_M=max(0.01, 2.0 / n))
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, **KKT)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**dict(st, max_iter=6)))
    o.solve()
    bt.iterate(6)
    for v in ("x", "y", "sigma_inv", "dphi"):
        assert rel(bt.vec(v), o.vec(v)) <= RTOL
    L = ob.lib()
    ln = ob.c_int(0)

    def oracle_ivec(name):
        ptr = L.oq_get_ivec(o.w, name.encode(), C.byref(ln))
        return np.ctypeslib.as_array(ptr, shape=(m,))

    # a synthetic active set, the same on both sides; form + factorise
    act = (np.arange(m) % 3 == 0).astype(np.int64)
    oracle_ivec("active")[:] = act
    bt.set_ivec("active", act)
    np_ = n + m
    bt.op("kkt_form")
    # the formed matrix (lower triangle of the slot, leading dimension ld) against numpy on the device's own scaled data
    nzA, nzQ = int(p.Ap[-1]), int(p.Qp[-1])
    A = sp.csc_matrix((bt.named_vec("A_values", nzA), p.Ai, p.Ap), shape=(m, n)).toarray()
    Ql = sp.csc_matrix((bt.named_vec("Q_values", nzQ), p.Qi, p.Qp), shape=(n, n)).toarray()
    K = np.eye(np_)
    K[:n, :n] = np.tril(Ql) + np.tril(Ql, -1).T + np.eye(n) / bt.stats(0).gamma
    sig_inv = bt.vec("sigma_inv")
    for k in np.where(act)[0]:
        K[n + k, :n] = A[k]
        K[:n, n + k] = A[k]
        K[n + k, n + k] = -sig_inv[k] if np.any(A[k]) else 1.0
    ld = -(-np_ // 8) * 8
    raw = bt.named_vec("L", ld * np_).reshape(np_, ld).T[:np_]      # [i, j] = entry (i, j) of the column-major slot
    assert np.array_equal(np.tril(raw), np.tril(K)) or np.max(np.abs(np.tril(raw) - np.tril(K))) <= 1e-15 * np.max(np.abs(K))
    bt.op("kkt_factorize")
    L.oq_kkt_form_and_factor(o.w)
    Lo, Do = o.kkt_factor()
    Lg, Dg = bt.factor_rows(np_)
    assert rel(Dg, Do) <= 1e-10 and np.max(np.abs(np.tril(Lg, -1) - np.tril(Lo, -1))) <= 1e-10
    # row additions for some inactive constraints, then row deletions for some active ones
    enter = np.where(act == 0)[0][:7]
    leave = np.where(act == 1)[0][2:7]
    for name, lst, cnt in (("enter", enter, "nb_enter"), ("leave", leave, "nb_leave")):
        bt.set_ivec(name, lst)
        bt.set_scalar(cnt, len(lst))
    act_new = act.copy(); act_new[enter] = 1; act_new[leave] = 0
    oracle_ivec("active_old")[:] = act
    oracle_ivec("active")[:] = act_new
    L.oq_set_entering_leaving_constraints(o.w)
    assert np.array_equal(o.ivec("enter"), enter) and np.array_equal(o.ivec("leave"), leave)
    bt.set_ivec("active", act_new)
    bt.op("kkt_update_entering_constraints")
    L.oq_kkt_update_entering_constraints(o.w)
    Lo, Do = o.kkt_factor()
    Lg, Dg = bt.factor_rows(np_)
    assert rel(Dg, Do) <= 1e-9 and np.max(np.abs(np.tril(Lg, -1) - np.tril(Lo, -1))) <= 1e-9, "after ladel_row_add"
    bt.op("kkt_update_leaving_constraints")
    L.oq_kkt_update_leaving_constraints(o.w)
    Lo, Do = o.kkt_factor()
    Lg, Dg = bt.factor_rows(np_)
    assert rel(Dg, Do) <= 1e-9 and np.max(np.abs(np.tril(Lg, -1) - np.tril(Lo, -1))) <= 1e-9, "after ladel_row_del"
    # the updated factor is the factor of the KKT matrix of the new active set (deleted rows: -1/sigma stays out, unit pivot)
    K2 = np.eye(np_)
    K2[:n, :n] = K[:n, :n]
    for k in np.where(act_new)[0]:
        K2[n + k, :n] = A[k]
        K2[:n, n + k] = A[k]
        K2[n + k, n + k] = -sig_inv[k] if np.any(A[k]) else 1.0
    assert np.max(np.abs((Lg * Dg) @ Lg.T - K2)) <= 1e-9 * max(1.0, np.max(np.abs(K2)))
    # kkt_solve
    rhs = np.random.default_rng(9).standard_normal(n)
    o.vec("dphi", copy=False)[:] = rhs
    bt.set_vec("dphi", rhs)
    bt.op("kkt_solve")
    L.oq_kkt_solve(o.w)
    assert rel(bt.vec("d"), o.vec("d")) <= 1e-9
    sol = bt.named_vec("sol_kkt", np_)
    assert rel(sol, o.vec("sol_kkt")) <= 1e-9
    assert np.max(np.abs(K2 @ sol - np.concatenate([-rhs, np.zeros(m)]))) <= 1e-8 * max(1.0, np.max(np.abs(rhs)))


def test_kkt_operations_are_refused_in_schur_mode(ctx):
    p = random_qp(12, 20, seed=5, density_A=0.3, density_M=0.2)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(verbose=0, factorization_method=1))
    from qpalm_amd.capi import QpgError
    with pytest.raises(QpgError):
        bt.op("kkt_form")


@pytest.mark.gpu
def test_config2_in_kkt_mode():
    """BASELINE.json config 2's QP (n = 1000, m = 2000) with FACTORIZE_KKT: a 3000-row quasi-definite panel (the large-factor sweep,
    k_solve<0>), row additions / deletions as the active set moves, iterative refinement -- against the oracle in KKT mode
    (24 s of CPU) and against this engine's Schur mode."""
    from qpalm_amd.solver import Context
    ctx = Context(0)
    ctx.set_option("coop", 0)
    p = random_qp(1000, 2000, seed=1000, density_A=0.01, density_M=0.005)
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    bk = QpalmBatch(ctx, [p], ctx.default_settings(**dict(st, **KKT)))
    bk.solve()
    bs = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    bs.solve()
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**dict(st, **KKT)))
    o.solve()
    info = bk.info(0)
    assert int(info.status_val) == o.status_val == 1
    assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
    (xk, yk), (xs, ys) = bk.solution(), bs.solution()
    assert rel(xk[0], o.x) <= 1e-8 and rel(yk[0], o.y) <= 1e-8
    assert rel(xk[0], xs[0]) <= 1e-5 and rel(yk[0], ys[0]) <= 1e-5      # two factorisation methods, one solution (to the tolerance of the solve)
    assert np.array_equal(bk.ivec("active", 0), o.ivec("active"))


def test_kkt_compact_factorisation_equals_the_full_one(ctx):
    """KKT mode's form + factorise on the variables and the ACTIVE constraints only (qpalm_kkt.h: kkt_form_compact / kkt_expand,
    round 4) against the dense factorisation of the whole (n+m) x (n+m) panel with its unit rows (ctx option kkt_compact = 0) and
    against the oracle's KKT factor: entry by entry after a few iterations (a refactorisation with a non-trivial active set, then
    row additions / deletions on the spread-out factor), and whole solves with identical counts and iterates."""
    n, m = sizes(ctx, (40, 70), (160, 270))
    p = random_qp(n, m, seed=77, density_A=max(0.02, 4.0 / n), density_M=max(0.01, 2.0 / n))
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, factorization_method=0)
    res = {}
    try:
        for comp in (1, 0):
            ctx.set_option("kkt_compact", comp)
            bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
            bt.iterate(4)
            Lm, D = bt.factor_rows(n + m)
            act = bt.ivec("active")
            bt.solve()
            x, y = bt.solution()
            res[comp] = (Lm, D, act, x[0].copy(), y[0].copy(), int(bt.info(0).iter), int(bt.stats(0).n_refactor))
            bt.close()
    finally:
        ctx.set_option("kkt_compact", 1)
    (L1, D1, a1, x1, y1, it1, rf1), (L0, D0, a0, x0, y0, it0, rf0) = res[1], res[0]
    assert np.array_equal(a1, a0) and 0 < int(a1.sum()) < m          # some constraints active, some not
    assert np.max(np.abs(D1 - D0)) <= 1e-12 * np.max(np.abs(D0)) and np.max(np.abs(np.tril(L1, -1) - np.tril(L0, -1))) <= 1e-12
    assert (it1, rf1) == (it0, rf0) and rel(x1, x0) <= 1e-12 and rel(y1, y0) <= 1e-10
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**dict(st, max_iter=4)))
    o.solve()
    Lo, Do = o.kkt_factor()
    assert rel(D1, Do) <= 1e-9 and np.max(np.abs(np.tril(L1, -1) - np.tril(Lo, -1))) <= 1e-9
