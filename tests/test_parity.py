"""Parity of the gfx950 path with the CPU oracle, through the C ABI (include/qpalm_gfx950.h).

Every test runs twice: `[emu]` executes the kernel SOURCE on the host (fiber emulation, CPU-only
test harness) and `[hip]` (marked gpu) executes the shipped HIP library on a real MI355X.

Tolerances (fp64): element-wise arithmetic is reproduced operation by operation, so differences
come only from reduction order / FMA inside SpMV, dot products and the LDL^T kernels:
  * final x, y and per-iteration iterates: <= 1e-9 relative (north-star: iterates to a stated fp64
    tolerance),  * final residual norms within 1e-8 of the oracle's,
  * active sets, entering/leaving lists, iteration counts and statuses: exact.
"""
import numpy as np
import scipy.sparse as sp
import pytest

from oracle import binding as ob
from qpalm_amd.problems import fixture_qp, random_mpc_qp, random_qp
from qpalm_amd.solver import Qpalm, QpalmBatch
from tests.helpers import STATUS

RTOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.size == 0 and b.size == 0:
        return 0.0
    return np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))


def oracle_for(p, st):
    o = ob.OracleQP(*p.args(), c=p.c, settings=ob.default_settings(**st))
    return o


def sizes(ctx, small, big):
    return small if ctx.kind == "emu" else big


# ---------------------------------------------------------------- the reference's own suites
def gsettings(ctx, golden, name, **over):
    st = dict(golden["expect"][name].get("settings", {}))
    st.update(over)
    st["verbose"] = 0
    return st


@pytest.mark.parametrize("name,over", [
    ("basic_qp", dict()), ("basic_qp", dict(scaling=0)), ("basic_qp", dict(proximal=0, scaling=2)),
    ("basic_qp", dict(proximal=0, scaling=0)), ("basic_qp", dict(sigma_max=1e3)),
    ("medium_qp", dict()), ("degen_hess", dict()), ("ls_qp", dict()),
])
def test_reference_solutions(ctx, golden, name, over):
    e = golden["expect"][name]
    st = gsettings(ctx, golden, name, **over)
    p = fixture_qp(golden["problems"][name])
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    bt.solve()
    info = bt.info(0)
    assert int(info.status_val) == STATUS["SOLVED"]
    x = bt.solution()[0][0]
    if "rel_tol" in e:
        for a, b in zip(x, e["solution"]):
            assert abs(a - b) <= abs(e["rel_tol"] * b)
    else:
        assert np.max(np.abs(x - e["solution"])) <= e["abs_tol"]
    o = oracle_for(p, st)
    o.solve()
    assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
    assert rel(x, o.x) <= RTOL and rel(bt.solution()[1][0], o.y) <= RTOL
    assert abs(info.pri_res_norm - o.info.pri_res_norm) <= 1e-8 and abs(info.dua_res_norm - o.info.dua_res_norm) <= 1e-8
    assert abs(info.objective - o.info.objective) <= 1e-9 * max(1.0, abs(o.info.objective))


@pytest.mark.parametrize("name,status", [("prim_inf_qp", "PRIMAL_INFEASIBLE"), ("dua_inf_qp", "DUAL_INFEASIBLE")])
@pytest.mark.parametrize("k", range(4))
def test_reference_infeasible(ctx, golden, name, status, k):
    st = gsettings(ctx, golden, name, **golden["expect"][name]["variants"][k])
    bt = QpalmBatch(ctx, [fixture_qp(golden["problems"][name])], ctx.default_settings(**st))
    bt.solve()
    assert int(bt.info(0).status_val) == STATUS[status]
    # ... in the oracle's iteration; dua_inf_qp variant 0 is the one exception: its certificate test is an equality in exact
    # arithmetic (test_dua_inf_decision_is_a_rounding_tie proves the tie with rational arithmetic)
    o = oracle_for(fixture_qp(golden["problems"][name]), st)
    o.solve()
    assert o.status_val == STATUS[status]
    if not (name == "dua_inf_qp" and k == 0):
        assert int(bt.info(0).iter) == int(o.info.iter), (int(bt.info(0).iter), int(o.info.iter))


def test_reference_status_paths(ctx, golden):
    p = fixture_qp(golden["problems"]["basic_qp"])
    base = gsettings(ctx, golden, "basic_qp")
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**dict(base, max_iter=1)))       # test_basic_qp_maxiter
    bt.solve()
    assert int(bt.info(0).status_val) == STATUS["MAX_ITER_REACHED"]
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**dict(base, eps_abs=1e-8, eps_rel=1e-8, inner_max_iter=2, max_iter=10)))
    bt.solve()                                                                       # test_basic_qp_inner_maxiter
    assert int(bt.info(0).status_val) == STATUS["MAX_ITER_REACHED"]
    assert int(bt.info(0).iter) == 10
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**dict(base, time_limit=0.01 * 1e-3)))  # test_basic_qp_time_limit
    bt.solve()
    assert int(bt.info(0).status_val) == STATUS["TIME_LIMIT_REACHED"]


@pytest.mark.parametrize("shape", [(70, 100), (200, 400), (200, 1300), (300, 500), (1000, 2000)])
def test_dual_objective_at_size(ctx, shape):
    """compute_dual_objective (iteration.c:272-299) on QPs that fill several wavefronts and both kernel instances, with
    and without scaling: status, iteration counts and the dual objective itself against the oracle.  (Round 3: the value
    was wrong on the hardware from about 70 variables on -- a compiler fault, tools/evidence/scan_exec_prologue.py -- while every
    small reference fixture passed.)"""
    n, m = shape
    if ctx.kind == "emu" and n > 100:
        pytest.skip("the emulator covers the smallest shape")
    p = random_qp(n, m, seed=1000, density_A=0.01 if n >= 400 else 4.0 / n, density_M=0.005 if n >= 400 else 2.0 / n)
    for extra in (dict(scaling=0), dict(), dict(dual_objective_limit=-1e3)):
        st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, enable_dual_termination=1, **extra)
        bt = QpalmBatch(ctx, [p, p], ctx.default_settings(**st))
        bt.solve()
        o = oracle_for(p, st)
        o.solve()
        for k in range(2):
            info = bt.info(k)
            assert int(info.status_val) == o.status_val
            assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
            assert abs(info.dual_objective - o.info.dual_objective) <= 1e-9 * max(1.0, abs(o.info.dual_objective))
            assert abs(info.objective - o.info.objective) <= 1e-9 * max(1.0, abs(o.info.objective))


def test_reference_dual_objective(ctx, golden):
    """test_basic_qp_dual_objective / _dual_early_termination (tests/src/test_basic_qp.c:334-362): the second resident
    factor LD_Q, compute_dual_objective on the device and the DUAL_TERMINATED exit (qpalm.c:459-468,545-583)."""
    e = golden["expect"]["basic_qp"]
    p = fixture_qp(golden["problems"]["basic_qp"])
    st = gsettings(ctx, golden, "basic_qp", enable_dual_termination=1)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    bt.solve()
    info = bt.info(0)
    assert int(info.status_val) == STATUS["SOLVED"]
    for a, b in zip(bt.solution()[0][0], e["solution"]):
        assert abs(a - b) <= abs(e["rel_tol"] * b)
    assert abs(info.objective - info.dual_objective) <= e["dual_objective_tol"]
    o = oracle_for(p, st)
    o.solve()
    assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
    assert abs(info.dual_objective - o.info.dual_objective) <= 1e-9 * max(1.0, abs(o.info.dual_objective))
    # early termination: the limit is exceeded at the first outer update
    st2 = dict(st, dual_objective_limit=e["dual_limit_early"])
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st2))
    bt.solve()
    info = bt.info(0)
    assert int(info.status_val) == STATUS["DUAL_TERMINATED"] and int(info.iter_out) == 0
    o = oracle_for(p, st2)
    o.solve()
    assert o.status_val == STATUS["DUAL_TERMINATED"] and int(info.iter) == int(o.info.iter)
    assert abs(info.dual_objective - o.info.dual_objective) <= 1e-9 * max(1.0, abs(o.info.dual_objective))
    assert rel(bt.solution()[0][0], o.x) <= RTOL and rel(bt.solution()[1][0], o.y) <= RTOL
    # a batch whose members stop for different reasons + dual termination switched on by update_settings
    q = random_qp(40, 80, seed=11, density_A=0.1, density_M=0.08)
    stq = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    bt = QpalmBatch(ctx, [q, q], ctx.default_settings(**stq))
    bt.solve()
    assert bt.info(0).dual_objective == 0.0                                          # QPALM_NULL (B8)
    assert bt.update_settings(ctx.default_settings(**dict(stq, enable_dual_termination=1))) == 0
    bt.warm_start(None, None)
    bt.solve()
    o = oracle_for(q, dict(stq, enable_dual_termination=1))
    o.solve()
    for b in range(2):
        info = bt.info(b)
        assert int(info.status_val) == o.status_val == STATUS["SOLVED"] and int(info.iter) == int(o.info.iter)
        assert abs(info.dual_objective - o.info.dual_objective) <= 1e-9 * max(1.0, abs(o.info.dual_objective))
        assert abs(info.objective - info.dual_objective) <= 1e-4 * max(1.0, abs(info.objective))


@pytest.mark.parametrize("over,ykey", [(dict(scaling=2, proximal=1), "warm_y_scaled"), (dict(scaling=0, proximal=1), "warm_y"),
                                       (dict(scaling=2, proximal=0), "warm_y"), (dict(scaling=0, proximal=0), "warm_y")])
def test_reference_warm_start(ctx, golden, over, ykey):
    e = golden["expect"]["basic_qp"]
    st = gsettings(ctx, golden, "basic_qp", warm_start=1, **over)
    p = fixture_qp(golden["problems"]["basic_qp"])
    q = Qpalm(ctx)
    q.settings = ctx.default_settings(**st)
    q.set_problem(p)
    q.setup()
    q.warm_start(e["warm_x"], e[ykey])
    q.solve()
    assert int(q.info.iter) < e["warm_iter_lt"] and q.status_val == STATUS["SOLVED"]
    for a, b in zip(q.x, e["solution"]):
        assert abs(a - b) <= abs(1e-5 * b)
    o = oracle_for(p, st)
    o.warm_start(e["warm_x"], e[ykey])
    o.solve()
    assert int(q.info.iter) == int(o.info.iter)
    assert rel(q.x, o.x) <= RTOL and rel(q.y, o.y) <= RTOL


def test_reference_resolve_is_reproducible(ctx, golden):
    # test_basic_qp_warm_start_resolve (test_basic_qp.c:275-307): same iterates to 1e-15, same iter
    st = gsettings(ctx, golden, "basic_qp")
    q = Qpalm(ctx)
    q.settings = ctx.default_settings(**st)
    q.set_problem(fixture_qp(golden["problems"]["basic_qp"]))
    q.setup()
    x0, y0 = q.batch.vec("x"), q.batch.vec("y")
    q.solve()
    xs, ys, it = q.x.copy(), q.y.copy(), int(q.info.iter)
    q.warm_start(x0, y0)
    q.solve()
    assert int(q.info.iter) == it
    assert np.max(np.abs(q.x - xs)) <= 1e-15 and np.max(np.abs(q.y - ys)) <= 1e-15


def test_reference_update_suite(ctx, golden):
    # suite_update: three tests on ONE workspace, in order (test_update.c:91-148)
    e, pr = golden["expect"]["update"], golden["problems"]["update"]
    st = gsettings(ctx, golden, "update")
    p = fixture_qp(pr)
    q = Qpalm(ctx)
    q.settings = ctx.default_settings(**st)
    q.set_problem(p)
    q.setup()
    o = oracle_for(p, st)
    q.solve(); o.solve()
    assert q.status_val == STATUS["SOLVED"] and np.max(np.abs(q.x - e["first"])) <= 1e-5
    assert rel(q.x, o.x) <= RTOL
    s = ctx.default_settings(**st)
    so = ob.default_settings(**st)
    for t in (s, so):
        t.gamma_init *= 0.1; t.theta = 0.9; t.proximal = 1; t.scaling = 10
    assert q.update_settings(s) == 0
    o.update_settings(so)
    q.solve(); o.solve()
    assert q.status_val == STATUS["SOLVED"] and np.max(np.abs(q.x - e["first"])) <= 1e-5
    assert int(q.info.iter) == int(o.info.iter) and rel(q.x, o.x) <= RTOL
    bmin, bmax = np.array(pr["bmin"]), np.array(pr["bmax"])
    bmin[0], bmax[1] = e["new_bmin0"], e["new_bmax1"]
    assert q.update_bounds(bmin, bmax) == 0
    o.update_bounds(bmin, bmax)
    q.solve(); o.solve()
    assert q.status_val == STATUS["SOLVED"] and np.max(np.abs(q.x - e["after_bounds"])) <= 1e-5
    assert int(q.info.iter) == int(o.info.iter) and rel(q.x, o.x) <= RTOL
    q.update_bounds(pr["bmin"], pr["bmax"]); o.update_bounds(pr["bmin"], pr["bmax"])
    q.update_q(e["new_q"]); o.update_q(e["new_q"])
    q.solve(); o.solve()
    assert q.status_val == STATUS["SOLVED"] and np.max(np.abs(q.x - e["after_q"])) <= 1e-5
    assert int(q.info.iter) == int(o.info.iter) and rel(q.x, o.x) <= RTOL


def test_reference_error_handling(ctx, golden):
    from qpalm_amd.capi import QpgError
    pr = golden["problems"]["error_handling"]
    p = fixture_qp(pr)
    with pytest.raises(QpgError):   # test_invalid_settings_during_setup
        QpalmBatch(ctx, [p], ctx.default_settings(max_iter=-1))
    bad = fixture_qp(pr)
    bad.bmin[0], bad.bmax[0] = 5.0, 0.0
    with pytest.raises(QpgError):   # test_invalid_data_during_setup
        QpalmBatch(ctx, [bad], ctx.default_settings(verbose=0))
    bt = QpalmBatch(ctx, [p], ctx.default_settings(verbose=0))
    assert int(bt.info(0).status_val) == STATUS["UNSOLVED"]
    assert bt.update_settings(ctx.default_settings(max_iter=-10)) != 0
    assert int(bt.info(0).status_val) == STATUS["ERROR"]
    bt = QpalmBatch(ctx, [p], ctx.default_settings(verbose=0))
    assert bt.update_settings(ctx.default_settings(verbose=0, scaling=0)) != 0   # decreasing scaling
    assert int(bt.info(0).status_val) == STATUS["ERROR"]
    bt = QpalmBatch(ctx, [p], ctx.default_settings(verbose=0))
    assert bt.update_bounds(bad.bmin[None, :], bad.bmax[None, :]) != 0
    assert int(bt.info(0).status_val) == STATUS["ERROR"]


# ---------------------------------------------------------------- the boundary (solver_interface.h)
def test_boundary_golden_vectors(ctx, golden):
    e, pr = golden["expect"]["solver_interface"], golden["problems"]["solver_interface"]
    p = fixture_qp(pr)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(verbose=0, eps_abs=1e-6, eps_rel=1e-6, scaling=0))
    tol = e["tol"]
    assert np.max(np.abs(bt.mat_vec("A", e["x"]) - e["A_x"])) <= tol
    assert np.max(np.abs(bt.mat_vec("Q", e["x"]) - e["Q_x"])) <= tol
    assert np.max(np.abs(bt.mat_tpose_vec("Q", e["x"]) - e["Q_x"])) <= tol
    assert np.max(np.abs(bt.mat_tpose_vec("A", e["Ad_in"]) - e["At_Ad"])) <= tol
    # test_ldlchol (test_solver_interface.c:144-160)
    bt.set_scalar("proximal", 0)
    bt.set_vec("dphi", [-v for v in e["ldl_rhs"]])
    bt.op("ldlchol"); bt.op("ldlsolveLD_neg_dphi")
    assert np.max(np.abs(bt.vec("d") - e["ldl_d"])) <= tol
    bt.set_scalar("proximal", 1)
    bt.set_scalar("gamma", e["ldl_gamma"])
    bt.op("ldlchol"); bt.op("ldlsolveLD_neg_dphi")
    assert np.max(np.abs(bt.vec("d") - e["ldl_d_prox"])) <= tol


def _prep_pair(ctx, n, m, seed, **kw):
    p = random_qp(n, m, seed=seed, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n))
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    st.update(kw)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    o = oracle_for(p, st)
    return p, st, bt, o


def test_boundary_factor_update_solve_linesearch(ctx):
    """ldlcholQAtsigmaA, ldlupdate/ldldowndate, ldlsolveLD_neg_dphi and exact_linesearch on a state
    taken from the middle of an oracle solve: the factor itself is compared, entry by entry."""
    n, m = sizes(ctx, (70, 140), (300, 600))
    p, st, bt, o = _prep_pair(ctx, n, m, 1234)
    # run both for a few iterations so that sigma / iterates are non-trivial
    o.enable_trace(64)
    so = ob.default_settings(**dict(st, max_iter=6))
    o2 = ob.OracleQP(*p.args(), settings=so)
    o2.solve()
    bt.iterate(6)
    for v in ("x", "y", "Ax", "Qx"):
        assert rel(bt.vec(v), o2.vec(v)) <= RTOL
    L = ob.lib()
    # a synthetic active set: every third constraint
    act = (np.arange(m) % 3 == 0).astype(np.int64)
    o2.ivec("active")  # touch
    import ctypes as C
    ln = ob.c_int(0)
    pa = L.oq_get_ivec(o2.w, b"active", C.byref(ln))
    np.ctypeslib.as_array(pa, shape=(m,))[:] = act
    bt.set_ivec("active", act)
    L.oq_ldlcholQAtsigmaA(o2.w)
    bt.op("ldlcholQAtsigmaA")
    Lo, Do = o2.factor()
    Lg, Dg = bt.factor()
    assert rel(Dg, Do) <= 1e-11 and np.max(np.abs(np.tril(Lg, -1) - np.tril(Lo, -1))) <= 1e-10
    # rank update with entering = some inactive rows, downdate with leaving = some active rows
    enter = np.where(act == 0)[0][:21]
    leave = np.where(act == 1)[0][5:17]
    for name, lst in (("enter", enter), ("leave", leave)):
        bt.set_ivec(name, lst)
        pe = L.oq_get_ivec(o2.w, name.encode(), C.byref(ln))
        # oracle lists live in m-sized buffers
        np.ctypeslib.as_array(pe, shape=(m,))[:len(lst)] = lst
    # oracle counters nb_enter/nb_leave are set through the same lists' lengths
    # (oq_set_scalar has no entry for them; emulate set_entering_leaving by writing active_old)
    act_new = act.copy(); act_new[enter] = 1; act_new[leave] = 0
    pao = L.oq_get_ivec(o2.w, b"active_old", C.byref(ln))
    np.ctypeslib.as_array(pao, shape=(m,))[:] = act
    np.ctypeslib.as_array(pa, shape=(m,))[:] = act_new
    L.oq_set_entering_leaving_constraints(o2.w)
    assert np.array_equal(o2.ivec("enter"), enter) and np.array_equal(o2.ivec("leave"), leave)
    L.oq_ldlupdate_entering_constraints(o2.w)
    L.oq_ldldowndate_leaving_constraints(o2.w)
    bt.set_scalar("nb_enter", len(enter)); bt.set_scalar("nb_leave", len(leave))
    bt.op("ldlupdate_entering_constraints")
    bt.op("ldldowndate_leaving_constraints")
    Lo, Do = o2.factor()
    Lg, Dg = bt.factor()
    assert rel(Dg, Do) <= 1e-10 and np.max(np.abs(np.tril(Lg, -1) - np.tril(Lo, -1))) <= 1e-9
    # the updated factor must equal a fresh factorisation of the new active set (property)
    bt.set_ivec("active", act_new)
    np.ctypeslib.as_array(pa, shape=(m,))[:] = act_new
    L.oq_ldlcholQAtsigmaA(o2.w)
    Lf, Df = o2.factor()
    assert rel(Dg, Df) <= 1e-8
    # solve with the factor
    rhs = np.random.default_rng(5).standard_normal(n)
    o2.vec("dphi", copy=False)[:] = rhs
    bt.set_vec("dphi", rhs)
    bt.op("ldlcholQAtsigmaA")
    L.oq_ldlsolveLD_neg_dphi(o2.w)
    bt.op("ldlsolveLD_neg_dphi")
    assert rel(bt.vec("d"), o2.vec("d")) <= 1e-9
    # line search on that direction
    tau_o = L.oq_exact_linesearch(o2.w)
    tau_g = bt.exact_linesearch()
    assert abs(tau_g - tau_o) <= 1e-9 * max(1.0, abs(tau_o))
    assert rel(bt.vec("Qd"), o2.vec("Qd")) <= 1e-12 and rel(bt.vec("Ad"), o2.vec("Ad")) <= 1e-12
    assert np.array_equal(bt.vec("delta"), o2.vec("delta")) or rel(bt.vec("delta"), o2.vec("delta")) <= 1e-12


def test_boundary_ldlupdate_sigma_changed(ctx):
    """qpg_ldlupdate_sigma_changed against the oracle's ldlupdate_sigma_changed (solver_interface.c:443-503) on the
    same factor, the same changed-row list and the same At_scale: updated factor entry by entry, At_scale and
    At_sqrt_sigma restored as the reference leaves them."""
    import ctypes as C
    n, m = sizes(ctx, (70, 140), (300, 600))
    p, st, bt, o = _prep_pair(ctx, n, m, 4321)
    o2 = ob.OracleQP(*p.args(), settings=ob.default_settings(**dict(st, max_iter=6)))
    o2.solve()
    bt.iterate(6)
    L = ob.lib()
    ln = ob.c_int(0)
    act = (np.arange(m) % 4 != 1).astype(np.int64)
    pa = L.oq_get_ivec(o2.w, b"active", C.byref(ln))
    np.ctypeslib.as_array(pa, shape=(m,))[:] = act
    bt.set_ivec("active", act)
    L.oq_ldlcholQAtsigmaA(o2.w)
    bt.op("ldlcholQAtsigmaA")
    # 19 changed rows (> one 16-rank sweep), growth factors like update_sigma produces (sqrt(mult_factor) >= 1)
    rng = np.random.default_rng(9)
    changed = np.sort(rng.choice(np.where(act == 1)[0], size=19, replace=False))
    scale = np.ones(m)
    scale[changed] = np.sqrt(1.0 + 99.0 * rng.random(19))
    # ... one of them a sigma that grew by one unit in the last place: sqrt(mult_factor) rounds to 1.0, the row's update vector is exactly
    # zero and the reference's CHOLMOD branch scales the zeroed row back by 1/0 (NaN); engine and oracle rebuild the row from A'
    # (fresh-seed fuzz campaign of round 4, seed 204 case 155)
    scale[changed[7]] = 1.0
    o2.vec("At_scale", copy=False)[:] = scale
    bt.set_vec("At_scale", scale)
    pe = L.oq_get_ivec(o2.w, b"enter", C.byref(ln))
    np.ctypeslib.as_array(pe, shape=(m,))[:19] = changed
    bt.set_ivec("enter", changed)
    L.oq_set_scalar(o2.w, b"nb_sigma_changed", 19.0)
    bt.set_scalar("nb_sigma_changed", 19)
    Atss0 = bt.named_vec("At_sqrt_sigma", int(p.Ap[-1]))
    L.oq_ldlupdate_sigma_changed(o2.w)
    bt.op("ldlupdate_sigma_changed")
    Lo, Do = o2.factor()
    Lg, Dg = bt.factor()
    assert rel(Dg, Do) <= 1e-10 and np.max(np.abs(np.tril(Lg, -1) - np.tril(Lo, -1))) <= 1e-9
    assert np.array_equal(bt.vec("At_scale"), o2.vec("At_scale"))           # 1 / sqrt(1 - 1/s^2) on the changed rows (inf on the one-ulp row)
    assert np.isinf(bt.vec("At_scale")[changed[7]])
    Atss1 = bt.named_vec("At_sqrt_sigma", int(p.Ap[-1]))
    assert np.all(np.isfinite(Atss1)) and rel(Atss1, Atss0) <= 1e-15          # scaled by s and back by 1/s; the one-ulp row rebuilt
    assert np.all(np.isfinite(Lg)) and np.all(np.isfinite(Dg))
    # property: D grew (a positive semidefinite term was added)
    bt.op("ldlcholQAtsigmaA")
    assert np.all(Dg >= bt.factor()[1] * (1 - 1e-12))


def test_boost_gamma_at_size(ctx, golden):
    """boost_gamma + the Gershgorin pass of form_schur (iteration.c:158-211, nonconvex.c:185-210) at a size where the
    bound spans several wavefronts: block-diagonal QPs made of perturbed copies of the reference's basic_qp (the
    fixture on which the convex branch of qpalm.c:612-630 reaches boost_gamma; random / MPC QPs never do)."""
    from qpalm_amd.problems import replicated_qp
    base = fixture_qp(golden["problems"]["basic_qp"])
    st = gsettings(ctx, golden, "basic_qp")
    copies = sizes(ctx, 15, 60)                    # n = 60 / 240, m = 75 / 300
    nboost = 0
    for seed, pert in ((5, 0.01), (6, 0.1)):
        p = replicated_qp(base, copies, seed=seed, pert=pert)
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
        o = oracle_for(p, st)
        bt.solve(); o.solve()
        info, s = bt.info(0), bt.stats(0)
        assert int(info.status_val) == o.status_val == STATUS["SOLVED"]
        assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out)
        assert int(s.n_boost_gamma) == o.counter("n_boost_gamma")
        assert abs(s.gamma - o.scalar("gamma")) <= 1e-9 * abs(o.scalar("gamma"))  # max(gamma_max, 1e14 / Gershgorin bound)
        assert rel(bt.solution()[0][0], o.x) <= RTOL and rel(bt.solution()[1][0], o.y) <= RTOL
        assert np.array_equal(bt.ivec("active"), o.ivec("active"))
        nboost += int(s.n_boost_gamma)
    assert nboost >= 2, "boost_gamma was not reached"


def test_dua_inf_decision_is_a_rounding_tie(ctx, golden):
    """dua_inf_qp, variant (proximal, scaling 2): this path declares dual infeasibility at iteration 3, the oracle at 7.
    Q = 1e-10 I and eps_dual_inf^2 = 1e-10 make the test `dx'Q dx <= c eps^2 ||D dx||^2` (termination.c:232-235) an
    EQUALITY in exact arithmetic, so the decision is a rounding tie.  Shown here with exact rational arithmetic on the
    fp64 state at the deciding iteration: |lhs - rhs| is below the rounding-error bound of the expression itself."""
    from fractions import Fraction as F
    p = fixture_qp(golden["problems"]["dua_inf_qp"])
    st = gsettings(ctx, golden, "dua_inf_qp", **golden["expect"]["dua_inf_qp"]["variants"][0])
    o = oracle_for(p, st)
    o.enable_trace(64)
    o.solve()
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    bt.solve()
    assert int(bt.info(0).status_val) == o.status_val == STATUS["DUAL_INFEASIBLE"]
    # replay this path up to (not including) its deciding iteration and evaluate both sides exactly
    it_dec = int(bt.info(0).iter)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    bt.iterate(it_dec)
    assert bt.num_unfinished() == 1
    c = F(bt.stats(0).sc_c)
    D = [F(v) for v in bt.vec("D")]
    dx = [F(a) - F(b) for a, b in zip(bt.vec("x"), bt.vec("x_prev"))]
    tau, gamma = F(bt.stats(0).tau), F(bt.stats(0).gamma)
    Qdx = [F(a) - tau / gamma * F(b) for a, b in zip(bt.vec("Qd"), bt.vec("d"))]  # Qd holds tau * (Q d + d / gamma)
    lhs = sum(a * b for a, b in zip(dx, Qdx))
    eps2 = F(float(st.get("eps_dual_inf", 1e-5))) ** 2
    rhs = c * eps2 * sum((a * b) ** 2 for a, b in zip(D, dx))
    assert rhs > 0
    # Qd is stored as tau (Q d + d / gamma) with |d / gamma| >> |Q d| (Q = 1e-10 c D^2), so forming Qd - tau/gamma d cancels:
    # the rounding-error bound of the expression the reference evaluates is 8 eps sum |dx_j| (|Qd_j| + |tau/gamma d_j|)
    bound = 8 * F(np.finfo(float).eps) * sum(abs(a) * (abs(F(qd)) + abs(tau / gamma * F(dd))) for a, qd, dd in zip(dx, bt.vec("Qd"), bt.vec("d")))
    assert abs(lhs - rhs) <= bound, (float(abs(lhs - rhs)), float(bound))
    assert bound <= F(1, 10 ** 9) * rhs  # the tie band itself is tiny: the two sides agree to 9 digits
    # and in exact arithmetic on the PROBLEM DATA the two sides are identical: c D (1e-10 I) D  vs  c 1e-10 D^2
    Qs = fixture_qp(golden["problems"]["dua_inf_qp"]).Qx
    assert all(F(v) == F(1e-10) for v in Qs)
    # iterates of the two paths agree to rounding until this path stops
    tr = o.trace()
    b2 = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    for k in range(it_dec):
        b2.iterate(1)
        assert rel(b2.vec("x"), tr["x"][k]) <= RTOL


def test_boundary_residuals_and_active_sets_bit_exact(ctx):
    n, m = sizes(ctx, (50, 100), (400, 800))
    p, st, bt, o = _prep_pair(ctx, n, m, 77)
    so = ob.default_settings(**dict(st, max_iter=4))
    o2 = ob.OracleQP(*p.args(), settings=so)
    o2.solve()
    bt.iterate(4)
    L = ob.lib()
    # copy the oracle's state bit for bit, then compare one compute_residuals + active-set pass
    for v in ("x", "y", "Ax", "Qx", "x0", "sigma", "sigma_inv"):
        bt.set_vec(v, o2.vec(v))
    bt.set_scalar("gamma", o2.scalar("gamma"))
    L.oq_compute_residuals(o2.w)
    bt.op("compute_residuals")
    for v in ("Axys", "z", "pri_res", "yh"):     # pure element-wise: bit exact
        assert np.array_equal(bt.vec(v), o2.vec(v)), v
    assert rel(bt.vec("Atyh"), o2.vec("Atyh")) <= 1e-13 and rel(bt.vec("dphi"), o2.vec("dphi")) <= 1e-13
    L.oq_set_active_constraints(o2.w)
    L.oq_set_entering_leaving_constraints(o2.w)
    bt.set_ivec("active_old", o2.ivec("active_old"))
    bt.op("set_active_constraints")
    assert np.array_equal(bt.ivec("active"), o2.ivec("active"))
    ne, nl = int(bt.stats(0).nb_enter), int(bt.stats(0).nb_leave)
    assert np.array_equal(bt.ivec("enter", length=ne), o2.ivec("enter"))
    assert np.array_equal(bt.ivec("leave", length=nl), o2.ivec("leave"))


# ---------------------------------------------------------------- whole solves vs the oracle
def _compare_solve(ctx, probs, st, rank_thr=-1):
    ctx.set_option("update_rank_threshold", rank_thr)
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**st))
    bt.solve()
    xs, ys = bt.solution()
    for k, p in enumerate(probs):
        o = oracle_for(p, st)
        o.solve()
        info = bt.info(k)
        assert int(info.status_val) == o.status_val
        assert int(info.iter) == int(o.info.iter) and int(info.iter_out) == int(o.info.iter_out), (k, info.iter, o.info.iter)
        assert rel(xs[k], o.x) <= RTOL and rel(ys[k], o.y) <= RTOL
        assert abs(info.pri_res_norm - o.info.pri_res_norm) <= 1e-8
        assert abs(info.dua_res_norm - o.info.dua_res_norm) <= 1e-8
        if rank_thr < 0:
            s = bt.stats(k)
            assert int(s.n_refactor) == o.counter("n_refactor") and int(s.n_rank1) == o.counter("n_rank1")
        assert np.array_equal(bt.ivec("active", k), o.ivec("active"))
    ctx.set_option("update_rank_threshold", -1)
    return bt


def test_random_qps_match_oracle(ctx):
    n, m = sizes(ctx, (40, 80), (200, 400))
    nb = sizes(ctx, 3, 8)
    probs = [random_qp(n, m, seed=1000 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(nb)]
    _compare_solve(ctx, probs, dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0))


def test_random_qps_unscaled_noprox(ctx):
    n, m = sizes(ctx, (30, 60), (150, 300))
    probs = [random_qp(n, m, seed=2000 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(2)]
    _compare_solve(ctx, probs, dict(eps_abs=1e-7, eps_rel=1e-7, verbose=0, scaling=0))
    _compare_solve(ctx, probs, dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, proximal=0, scaling=3))


def test_refactor_policy_changes_speed_not_results(ctx):
    """The own refactor-vs-update threshold (a speed policy) must not change iterates."""
    n, m = sizes(ctx, (40, 80), (200, 400))
    probs = [random_qp(n, m, seed=3000, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n))]
    _compare_solve(ctx, probs, dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0), rank_thr=2)


def test_workgroup_shapes_and_placement_do_not_change_results(ctx):
    """QPs whose factor has at most 256 rows run on 256-thread workgroups (four per CU) by default; the 512-thread
    instance, and every workgroup running its serial chains on wavefront 0, must give the same answers (all against
    the oracle, and the launch shape is what the library reports)."""
    n, m = sizes(ctx, (40, 80), (160, 270))
    probs = [random_qp(n, m, seed=5100 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(sizes(ctx, 2, 6))]
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    try:
        for small, place in ((1, 1), (0, 1), (1, 0), (0, 0), (0, 2), (2, 0), (2, 1)):   # place 2: rows owned by SIMD (512-thread instance)
            ctx.set_option("small_workgroups", small)                   # small 2: the 128-thread instance (seven workgroups per CU)
            ctx.set_option("place_panel_wave", place)
            bt = _compare_solve(ctx, probs, st)
            wgs, threads, lds = bt.launch_shape()
            assert wgs >= 1 and lds >= (16 if small == 2 else 32) * 1024
            if ctx.kind == "hip":
                assert threads == {0: 512, 1: 256, 2: 128}[small]
    finally:
        ctx.set_option("small_workgroups", 1)
        ctx.set_option("place_panel_wave", 0)


def test_large_batches_of_small_qps_run_on_the_128_thread_instance(ctx):
    """more QPs than the 256-thread instance keeps resident (1024), factors of at most 192 rows: the library picks the 128-thread
    instance (seven workgroups per CU: 1792 QPs in flight) by itself -- every QP of the batch against the oracle, through the work queue"""
    if ctx.kind != "hip":
        pytest.skip("the emulation build holds one instance of the kernels")
    distinct = [random_qp(48, 96, seed=5300 + k, density_A=0.08, density_M=0.04) for k in range(24)]
    probs = [distinct[k % len(distinct)] for k in range(2000)]
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**st))
    wgs, threads, lds = bt.launch_shape()
    assert threads == 128 and wgs == 1792, (wgs, threads, lds)
    bt.solve()
    xs, ys = bt.solution()
    ref = []
    for p in distinct:
        o = oracle_for(p, st)
        o.solve()
        ref.append((o.status_val, int(o.info.iter), o.x.copy(), o.y.copy()))
    for k in range(len(probs)):
        stv, it, x, y = ref[k % len(distinct)]
        info = bt.info(k)
        assert int(info.status_val) == stv and int(info.iter) == it, (k, int(info.iter), it)
        assert rel(xs[k], x) <= RTOL and rel(ys[k], y) <= RTOL
    bt.close()
    ctx.set_option("small_workgroups", 3)     # the same batch on the 256-thread instance
    try:
        bt = QpalmBatch(ctx, probs[:1100], ctx.default_settings(**st))
        assert bt.launch_shape()[1] == 256
        bt.close()
    finally:
        ctx.set_option("small_workgroups", 1)


def _with_long_row_and_column(p, seed):
    """p with one dense row of A (> 16 entries: several chunks of a quarter wavefront) and one dense column"""
    rng = np.random.default_rng(seed)
    A = p.A_mat().tolil()
    r, c = p.m // 3, p.n // 4
    cols = rng.choice(p.n, size=min(p.n, 37), replace=False)
    A[r, cols] = rng.standard_normal(len(cols))
    rows = rng.choice(p.m, size=min(p.m, 41), replace=False)
    A[rows, c] = rng.standard_normal(len(rows))
    A = A.tocsc(); A.sort_indices()
    return type(p)(p.n, p.m, p.Qp, p.Qi, p.Qx, A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64),
                   p.q, p.bmin, p.bmax)


def test_schur_assembly_variants_match_oracle(ctx):
    """form_schur_narrow (a quarter wavefront per column, small QPs with short rows on average) with rows and columns of
    A longer than one chunk of 16, against the oracle and against the one-wavefront-per-column assembly (bit for bit)."""
    n, m = sizes(ctx, (60, 120), (200, 400))
    probs = [_with_long_row_and_column(random_qp(n, m, seed=5300 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)), k)
             for k in range(sizes(ctx, 2, 4))]
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    sols = []
    try:
        for narrow in (1, 0):
            ctx.set_option("narrow_rows", narrow)
            bt = _compare_solve(ctx, probs, st)
            sols.append(bt.solution())
            # the assembled + factorised matrix itself, on a fresh active set
            bt.op("ldlcholQAtsigmaA")
            sols.append(bt.factor())
    finally:
        ctx.set_option("narrow_rows", 1)
    assert np.array_equal(sols[0][0], sols[2][0]) and np.array_equal(sols[0][1], sols[2][1])
    assert np.array_equal(sols[1][0], sols[3][0]) and np.array_equal(sols[1][1], sols[3][1])


def test_no_constraints_and_single_constraint(ctx):
    """Edge cases of the shapes: m = 0 (A is n x 0: the loop never has an active row; x = -Q^{-1} q) and m = 1."""
    base = random_qp(30, 10, seed=7, density_A=0.2, density_M=0.1)
    QPt = type(base)
    p0 = QPt(base.n, 0, base.Qp, base.Qi, base.Qx, np.zeros(base.n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64), np.zeros(0),
             base.q, np.zeros(0), np.zeros(0))
    row = np.zeros(base.n); row[[2, 7, 11]] = [1.0, -2.0, 0.5]
    A1 = sp.csc_matrix(row[None, :])
    p1 = QPt(base.n, 1, base.Qp, base.Qi, base.Qx, A1.indptr.astype(np.int64), A1.indices.astype(np.int64), A1.data.astype(np.float64),
             base.q, np.array([-0.05]), np.array([0.05]))
    st = dict(eps_abs=1e-8, eps_rel=1e-8, verbose=0)
    bt = _compare_solve(ctx, [p0], st)
    assert rel(bt.solution()[0][0], np.linalg.solve(p0.Q_full().toarray(), -p0.q)) <= 1e-9
    _compare_solve(ctx, [p1], st)


def test_replacing_a_member_and_setting_up_again(ctx):
    """qpg_batch_set_problem on a member that is already set (fewer nonzeros, other values) followed by a second qpg_batch_setup: the
    batch then solves exactly what a fresh batch of the new members solves (the host slab's padding behind the shorter arrays and the
    device arena are cleared)."""
    from qpalm_amd.capi import f64, fptr, i64, iptr
    dense = [random_qp(24, 30, seed=50 + k, density_A=0.5, density_M=0.4) for k in range(3)]
    thin = [random_qp(24, 30, seed=60 + k, density_A=0.1, density_M=0.05) for k in range(3)]
    st = dict(eps_abs=1e-8, eps_rel=1e-8, verbose=0)
    nzA, nzQ = max(int(p.Ap[-1]) for p in dense), max(int(p.Qp[-1]) for p in dense)
    assert all(int(p.Ap[-1]) < nzA and int(p.Qp[-1]) < nzQ for p in thin)
    bt = QpalmBatch(ctx, dense, ctx.default_settings(**st))
    bt.solve()
    for k, p in enumerate(thin):
        a = (i64(p.Qp), i64(p.Qi), f64(p.Qx), i64(p.Ap), i64(p.Ai), f64(p.Ax), f64(p.q), f64(p.bmin), f64(p.bmax))
        bt._check(bt.L.qpg_batch_set_problem(bt.h, k, iptr(a[0]), iptr(a[1]), fptr(a[2]), iptr(a[3]), iptr(a[4]), fptr(a[5]), fptr(a[6]), 0.0, fptr(a[7]), fptr(a[8])))
    bt._check(bt.L.qpg_batch_setup(bt.h))
    bt.solve()
    fresh = QpalmBatch(ctx, thin, ctx.default_settings(**st))
    fresh.solve()
    for got, want in zip(bt.solution(), fresh.solution()):
        assert np.array_equal(got, want)
    assert [i.iter for i in bt.infos()] == [i.iter for i in fresh.infos()]
    _compare_solve(ctx, thin, st)


def test_mpc_qps_match_oracle(ctx):
    T = sizes(ctx, 3, 10)
    probs = [random_mpc_qp(T=T, nx=sizes(ctx, 4, 10), nu=sizes(ctx, 2, 5), seed=k) for k in range(2)]
    _compare_solve(ctx, probs, dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0))


def test_per_iteration_trace(ctx):
    """Per-iteration iterates, step sizes and active sets against the oracle's trace."""
    n, m = sizes(ctx, (40, 80), (250, 500))
    p, st, bt, o = _prep_pair(ctx, n, m, 4242)
    o.enable_trace(400)
    o.solve()
    tr = o.trace()
    K = len(tr["kind"])
    assert K >= 8
    for k in range(K):
        bt.iterate(1)
        s = bt.stats(0)
        assert int(s.last_kind) == int(tr["kind"][k]), k
        assert rel(bt.vec("x"), tr["x"][k]) <= RTOL and rel(bt.vec("y"), tr["y"][k]) <= RTOL, k
        if tr["kind"][k] == 0:
            assert abs(s.tau - tr["tau"][k]) <= 1e-9 * max(1.0, abs(tr["tau"][k])), k
            assert int(s.last_fact) == int(tr["fact"][k]), k
            assert np.array_equal(bt.ivec("active"), tr["active"][k]), k
            assert (int(s.nb_active), int(s.nb_enter), int(s.nb_leave)) == (int(tr["nb_active"][k]), int(tr["nb_enter"][k]), int(tr["nb_leave"][k]))
            assert rel(bt.vec("d"), tr["d"][k]) <= 1e-8, k
        assert abs(s.gamma - tr["gamma"][k]) <= 1e-12 * abs(tr["gamma"][k]), k
    bt.iterate(1)
    assert bt.num_unfinished() == 0 and int(bt.info(0).status_val) == o.status_val and int(bt.info(0).iter) == int(o.info.iter)


def test_work_queue_more_qps_than_slots(ctx):
    """B > max_slots: workgroups pull QPs from the atomic queue and reuse their factor slot."""
    ctx.set_option("max_slots", 2)
    n, m = sizes(ctx, (24, 48), (100, 200))
    probs = [random_qp(n, m, seed=500 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(5)]
    try:
        _compare_solve(ctx, probs, dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
    finally:
        ctx.set_option("max_slots", 512)


def test_work_queue_starts_the_longest_previous_solves_first(ctx):
    """Round 6: a launch through the work queue hands the members out in descending order of the kernel time of their PREVIOUS solve
    (k_queue_order: the tail of the launch is then made of short solves; identity before the first solve).  The order must be exactly that
    permutation, and must not change any result: the second solve equals the first bit for bit, and both equal the oracle."""
    ctx.set_option("max_slots", 2)
    n, m = sizes(ctx, (24, 48), (300, 600))   # (on the hardware: the 512-thread instance, whose slot count is max_slots itself)
    probs = [random_qp(n, m, seed=520 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(7)]
    try:
        bt = _compare_solve(ctx, probs, dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0))
        B = len(probs)
        assert bt.launch_shape()[0] < B                                                    # the launch goes through the work queue
        assert [int(bt.ivec("queue_order", k, 1)[0]) for k in range(B)] == list(range(B))     # no previous solve: index order
        x1, y1 = [a.copy() for a in bt.solution()]
        cost = [float(bt.stats(k).ms_total) for k in range(B)]
        bt.warm_start(None, None)
        bt.solve()
        order = [int(bt.ivec("queue_order", k, 1)[0]) for k in range(B)]
        assert order == sorted(range(B), key=lambda k: (-cost[k], k)), (order, cost)
        x2, y2 = bt.solution()
        assert np.array_equal(x1, x2) and np.array_equal(y1, y2)
        bt.close()
    finally:
        ctx.set_option("max_slots", 512)


def test_warm_started_mpc_sequence(ctx):
    """update_bounds + warm_start between solves (the MPC use of the reference,
    simulations/randomMPCsequential.m:158-177)."""
    T, nx, nu = sizes(ctx, (3, 4, 2), (10, 10, 5))
    p = random_mpc_qp(T=T, nx=nx, nu=nu, seed=3)
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    q = Qpalm(ctx); q.settings = ctx.default_settings(**st); q.set_problem(p); q.setup()
    o = oracle_for(p, st)
    q.solve(); o.solve()
    rng = np.random.default_rng(0)
    for step in range(2):
        bmin, bmax = p.bmin.copy(), p.bmax.copy()
        x0 = 0.5 * rng.standard_normal(nx)
        bmin[:nx] = x0; bmax[:nx] = x0
        assert q.update_bounds(bmin, bmax) == 0
        o.update_bounds(bmin, bmax)
        xw, yw = q.x.copy(), q.y.copy()
        q.warm_start(xw, yw); o.warm_start(xw, yw)
        q.solve(); o.solve()
        assert q.status_val == o.status_val and int(q.info.iter) == int(o.info.iter)
        assert rel(q.x, o.x) <= RTOL and rel(q.y, o.y) <= RTOL


def test_large_factor_path(ctx):
    """Factors with more rows than 4 x workgroup size use the large-factor update sweep (running vectors in HBM,
    dense_updown_big / k_solve<0>): n = 530 on the emulator (workgroups of 128 threads there), n = 2500 on the GPU
    (workgroups of 512 threads: above the 2048-row limit of the register-resident sweep; BASELINE.json config 5 is n = 5000)."""
    n, m = sizes(ctx, (530, 700), (2500, 3200))
    p = random_qp(n, m, seed=11, density_A=sizes(ctx, 0.02, 0.004), density_M=sizes(ctx, 0.01, 0.002))
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    bt = _compare_solve(ctx, [p], st)
    assert int(bt.stats(0).n_rank1) > 0 and int(bt.stats(0).n_sweeps) > 0


def test_linesearch_sort_buffer_in_hbm_with_lds_tiles(ctx):
    """more breakpoints than the LDS holds (m > 3200 or so: configs 2' and 5, every large sparse QP) are sorted in HBM, a tile of them at a
    time through all the short-distance steps of the bitonic network in LDS.  Same network: forced here on small QPs (context option
    linesearch_hbm), the iterates must be BIT-identical to the in-LDS sort's, and equal to the oracle's."""
    n, m = sizes(ctx, (40, 90), (300, 700))
    probs = [random_qp(n, m, seed=8100 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(2)]
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    out = {}
    try:
        for hbm in (0, 1, 32):       # 32: tiles of 32 entries -- several tiles and the long-distance steps in HBM between them
            ctx.set_option("linesearch_hbm", hbm)
            bt = _compare_solve(ctx, probs, st)
            out[hbm] = bt.solution()
            bt.close()
    finally:
        ctx.set_option("linesearch_hbm", 0)
    for hbm in (1, 32):
        assert np.array_equal(out[0][0], out[hbm][0]) and np.array_equal(out[0][1], out[hbm][1]), hbm


def _with_Q(p, kind):
    """the problem with another Hessian: "lp" Q = 0; "tiny" Q = 1e-10 I; "rank1" Q = 1e-10 v v' (positive diagonal, rank one)"""
    n = p.n
    if kind == "lp":
        Qd = np.zeros((n, n))
    elif kind == "tiny":
        Qd = 1e-10 * np.eye(n)
    else:
        v = 1.0 + np.arange(n) / n
        Qd = 1e-10 * np.outer(v, v)
    Ql = sp.csc_matrix(np.tril(Qd)) if kind != "lp" else sp.csc_matrix((n, n))
    Ql.sort_indices()
    return type(p)(p.n, p.m, Ql.indptr.astype(np.int64), Ql.indices.astype(np.int64), Ql.data.astype(np.float64), p.Ap, p.Ai, p.Ax, p.q, p.bmin, p.bmax)


def test_downdate_into_a_numerically_singular_matrix_is_backward_stable(ctx):
    """Round 5, fuzz LP case 701 / 128 boiled down: H = Q + A' Sigma A + I / gamma with gamma = 1e7 and sigma = 1e3; 86 of the 90 active rows
    leave at once (six sweeps of 16 ranks), lambda_min goes from 150 to 1e-7.  With the sweep's pivots taken as d_0 + sum(p) the factor
    that comes out has a backward error of 2e-8; carried as the running pivot d_r = d_{r-1} + p_r it is backward stable.  Round 6: in the
    DEFAULT configuration the library sums a column as running pivots whenever a pivot shrinks by 2^8 or more inside a sweep (the per-column
    guard of qp_rank_pivots), whatever Q looks like: an LP (also flagged at setup: every column sequential), Q = 1e-10 I and a rank-one
    positive semidefinite Q with a positive diagonal (the two the structural hint of round 5 missed) must all stay below 1e-13; the
    unguarded tree (sequential_rank_sums = 0, an A/B option) is run beside them and loses eight digits."""
    n, m = 40, 120
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, scaling=0, gamma_init=1e7, gamma_max=1e7, sigma_init=1e3)
    err, nseq = {}, {}
    try:
        for mode, kind in ((0, "tiny"), (-1, "lp"), (-1, "tiny"), (-1, "rank1"), (1, "tiny")):
            ctx.set_option("sequential_rank_sums", mode)
            p = _with_Q(random_qp(n, m, seed=9100, density_A=0.08, density_M=0.05), kind)
            bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
            bt.begin_solve()
            bt.iterate(2)                                     # sigma and A' sqrt(Sigma) are set up by the first iterations
            A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n)).toarray()
            Ql = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(p.n, p.n)).toarray()
            Qd = Ql + np.tril(Ql, -1).T
            act = np.zeros(m, dtype=np.int64); act[:90] = 1
            bt.set_ivec("active", act)
            bt.op("ldlcholQAtsigmaA")
            leave = np.where(act == 1)[0][2:88]
            bt.set_ivec("leave", leave); bt.set_scalar("nb_leave", len(leave)); bt.set_scalar("nb_enter", 0)
            bt.op("ldldowndate_leaving_constraints")
            L, D = bt.factor(0)
            L = np.tril(L, -1) + np.eye(n)
            keep = act.copy(); keep[leave] = 0
            sig, gam = bt.vec("sigma", 0)[:m], float(bt.stats(0).gamma)
            H = Qd + (A[keep == 1].T * sig[keep == 1]) @ A[keep == 1] + np.eye(n) / gam
            H0 = Qd + (A[act == 1].T * sig[act == 1]) @ A[act == 1] + np.eye(n) / gam
            assert np.min(np.linalg.eigvalsh(H)) < 1e-8 * np.min(np.linalg.eigvalsh(H0))
            err[mode, kind] = np.max(np.abs(L @ np.diag(D) @ L.T - H)) / np.max(np.abs(H0))
            nseq[mode, kind] = int(bt.stats(0).n_seq_columns)
            bt.close()
    finally:
        ctx.set_option("sequential_rank_sums", -1)
    for key in ((-1, "lp"), (-1, "tiny"), (-1, "rank1"), (1, "tiny")):
        assert err[key] <= 1e-13, err
    assert err[0, "tiny"] >= 1e-10, err          # (what the unguarded prefix-tree form loses on this matrix: the reason for the guard)
    assert nseq[-1, "tiny"] > 0 and nseq[-1, "rank1"] > 0 and nseq[0, "tiny"] == 0 and nseq[1, "tiny"] == 0, nseq   # the guard is what did it
