"""Pins the CPU oracle against every golden vector the reference's own tests hold for the path.

Each test names the reference test it restates (tests/src/*.c).  The reference runs every
integration case three times (KKT_OR_SCHUR / KKT / SCHUR, e.g. test_basic_qp.c:410-427); the
CHOLMOD build forces SCHUR for all three (solver_interface.c:72-74), which is the path restated.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as ob
from tests.helpers import STATUS, prob_args


def mk(golden, name, **over):
    st = dict(golden["expect"][name].get("settings", {}))
    st.update(over)
    s = ob.default_settings(verbose=0, **st)
    return ob.OracleQP(*prob_args(golden["problems"][name]), settings=s), s


def assert_rel(x, sol, rel):
    for a, b in zip(x, sol):
        assert abs(a - b) <= abs(rel * b), (x, sol)


# ---------------------------------------------------------------- suite_basic_qp
def basic_settings(golden, **over):
    # basic_qp_test_setup, test_basic_qp.c:90-103
    base = dict(proximal=1, scaling=10, warm_start=0, max_iter=10000, inner_max_iter=100, eps_abs=1e-6,
                eps_rel=1e-6, enable_dual_termination=0, dual_objective_limit=1e20, sigma_max=1e9,
                time_limit=1e20, gamma_init=1e1, max_rank_update_fraction=1.0)
    base.update(over)
    return base


@pytest.mark.parametrize("over", [
    dict(),                               # test_basic_qp
    dict(scaling=0),                      # test_basic_qp_unscaled
    dict(proximal=0, scaling=2),          # test_basic_qp_noprox
    dict(proximal=0, scaling=0),          # test_basic_qp_noprox_unscaled
    dict(sigma_max=1e3),                  # test_basic_qp_sigma_max
])
def test_basic_qp(golden, over):
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden, **over))
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert_rel(qp.x, golden["expect"]["basic_qp"]["solution"], 1e-5)


@pytest.mark.parametrize("over,ykey", [
    (dict(scaling=2, proximal=1), "warm_y_scaled"),   # test_basic_qp_warm_start (:189-207)
    (dict(scaling=0, proximal=1), "warm_y"),          # _unscaled
    (dict(scaling=2, proximal=0), "warm_y"),          # _noprox
    (dict(scaling=0, proximal=0), "warm_y"),          # _noprox_unscaled
])
def test_basic_qp_warm_start(golden, over, ykey):
    e = golden["expect"]["basic_qp"]
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden, warm_start=1, **over))
    qp.warm_start(e["warm_x"], e[ykey])
    qp.solve()
    assert qp.info.iter < e["warm_iter_lt"]
    assert qp.status_val == STATUS["SOLVED"]
    assert_rel(qp.x, e["solution"], 1e-5)


def test_basic_qp_warm_start_resolve(golden):
    # test_basic_qp.c:275-307: identical iterates (1e-15) and iteration count on re-solve
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden))
    x0, y0 = qp.vec("x"), qp.vec("y")
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    xs, ys, it = qp.x, qp.y, int(qp.info.iter)
    qp.warm_start(x0, y0)
    qp.solve()
    assert int(qp.info.iter) == it
    assert np.max(np.abs(qp.x - xs)) <= 1e-15
    assert np.max(np.abs(qp.y - ys)) <= 1e-15


def test_basic_qp_maxiter(golden):
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden, max_iter=1))
    qp.solve()
    assert qp.status_val == STATUS["MAX_ITER_REACHED"]


def test_basic_qp_inner_maxiter(golden):
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden, eps_abs=1e-8, eps_rel=1e-8, inner_max_iter=2, max_iter=10))
    qp.solve()
    assert qp.status_val == STATUS["MAX_ITER_REACHED"]


def test_basic_qp_dual_objective(golden):
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden, enable_dual_termination=1))
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert_rel(qp.x, golden["expect"]["basic_qp"]["solution"], 1e-5)
    assert abs(qp.info.objective - qp.info.dual_objective) <= 1e-5


def test_basic_qp_dual_early_termination(golden):
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden, enable_dual_termination=1, dual_objective_limit=-1e9))
    qp.solve()
    assert qp.status_val == STATUS["DUAL_TERMINATED"]
    assert qp.info.iter_out == 0


def test_basic_qp_time_limit(golden):
    qp, _ = mk(golden, "basic_qp", **basic_settings(golden, time_limit=0.01 * 1e-3))
    qp.solve()
    assert qp.status_val == STATUS["TIME_LIMIT_REACHED"]


# ---------------------------------------------------------------- other integration suites
def test_medium_qp(golden):
    qp, _ = mk(golden, "medium_qp")
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert_rel(qp.x, golden["expect"]["medium_qp"]["solution"], 1e-5)


def test_degen_hess(golden):
    qp, _ = mk(golden, "degen_hess")
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert np.max(np.abs(qp.x - golden["expect"]["degen_hess"]["solution"])) <= 1e-5


def test_ls_qp(golden):
    qp, _ = mk(golden, "ls_qp")
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert np.max(np.abs(qp.x - golden["expect"]["ls_qp"]["solution"])) <= 1e-5


@pytest.mark.parametrize("k", range(4))
def test_prim_inf_qp(golden, k):
    qp, _ = mk(golden, "prim_inf_qp", **golden["expect"]["prim_inf_qp"]["variants"][k])
    qp.solve()
    assert qp.status_val == STATUS["PRIMAL_INFEASIBLE"]


@pytest.mark.parametrize("k", range(4))
def test_dua_inf_qp(golden, k):
    qp, _ = mk(golden, "dua_inf_qp", **golden["expect"]["dua_inf_qp"]["variants"][k])
    qp.solve()
    assert qp.status_val == STATUS["DUAL_INFEASIBLE"]


def test_update_suite(golden):
    # suite_update runs its three tests on ONE workspace in order (test_update.c:91-148)
    e = golden["expect"]["update"]
    p = golden["problems"]["update"]
    qp, s = mk(golden, "update")
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert np.max(np.abs(qp.x - e["first"])) <= 1e-5
    s.gamma_init *= 0.1
    s.theta = 0.9
    s.proximal = 1
    s.scaling = 10
    qp.update_settings(s)
    assert qp.status_val != STATUS["ERROR"]
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert np.max(np.abs(qp.x - e["first"])) <= 1e-5
    # test_update_bounds
    bmin, bmax = np.array(p["bmin"]), np.array(p["bmax"])
    bmin[0], bmax[1] = e["new_bmin0"], e["new_bmax1"]
    qp.update_bounds(bmin, bmax)
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert np.max(np.abs(qp.x - e["after_bounds"])) <= 1e-5
    qp.update_bounds(p["bmin"], p["bmax"])
    # test_update_q
    qp.update_q(e["new_q"])
    qp.solve()
    assert qp.status_val == STATUS["SOLVED"]
    assert np.max(np.abs(qp.x - e["after_q"])) <= 1e-5


def test_error_handling(golden):
    # test_error_handling.c:88-134
    p = golden["problems"]["error_handling"]
    s = ob.default_settings(verbose=0, max_iter=-1)
    assert not ob.OracleQP(*prob_args(p), settings=s).ok
    bad = dict(p)
    bad["bmin"] = [5.0] + p["bmin"][1:]
    bad["bmax"] = [0.0] + p["bmax"][1:]
    assert not ob.OracleQP(*prob_args(bad), settings=ob.default_settings(verbose=0)).ok
    s = ob.default_settings(verbose=0)
    qp = ob.OracleQP(*prob_args(p), settings=s)
    assert qp.status_val == STATUS["UNSOLVED"]
    s.max_iter = -10
    qp.update_settings(s)
    assert qp.status_val == STATUS["ERROR"]
    s = ob.default_settings(verbose=0)
    qp = ob.OracleQP(*prob_args(p), settings=s)
    s.scaling = 0
    qp.update_settings(s)
    assert qp.status_val == STATUS["ERROR"]
    qp = ob.OracleQP(*prob_args(p), settings=ob.default_settings(verbose=0))
    qp.update_bounds(bad["bmin"], bad["bmax"])
    assert qp.status_val == STATUS["ERROR"]


# ---------------------------------------------------------------- suite_solver (the boundary)
def test_solver_interface(golden):
    e = golden["expect"]["solver_interface"]
    p = golden["problems"]["solver_interface"]
    L = ob.lib()
    A, ka = ob.make_sparse(p["m"], p["n"], p["Ap"], p["Ai"], p["Ax"], 0)
    Q, kq = ob.make_sparse(p["n"], p["n"], p["Qp"], p["Qi"], p["Qx"], -1)
    tol = e["tol"]
    x = ob.f64(e["x"])
    y = np.zeros(3)
    L.oq_mat_vec(C.byref(A), ob.fptr(x), ob.fptr(y))                     # test_mat_vec :106-111
    assert np.max(np.abs(y - e["A_x"])) <= tol
    xq = ob.f64(e["x"])
    L.oq_mat_vec(C.byref(Q), ob.fptr(xq), ob.fptr(xq))                   # aliased, :112-115
    assert np.max(np.abs(xq - e["Q_x"])) <= tol
    xq = ob.f64(e["x"])
    L.oq_mat_tpose_vec(C.byref(Q), ob.fptr(xq), ob.fptr(xq))             # :117-119
    assert np.max(np.abs(xq - e["Q_x"])) <= tol
    ad = ob.f64(e["Ad_in"])
    out = np.zeros(2)
    L.oq_mat_tpose_vec(C.byref(A), ob.fptr(ad), ob.fptr(out))            # test_mat_tpose_vec :123-127
    assert np.max(np.abs(out - e["At_Ad"])) <= tol
    cols, rows = np.zeros(2), np.zeros(3)
    L.oq_mat_inf_norm_cols(C.byref(A), ob.fptr(cols))                    # :129-133
    L.oq_mat_inf_norm_rows(C.byref(A), ob.fptr(rows))                    # :135-140
    assert np.max(np.abs(cols - e["inf_norm_cols"])) <= tol
    assert np.max(np.abs(rows - e["inf_norm_rows"])) <= tol
    # test_ldlchol :144-160 (CHOLMOD only): factor the *user's* Q, not the scaled copy
    qp = ob.OracleQP(*prob_args(p), settings=ob.default_settings(verbose=0, eps_abs=1e-6, eps_rel=1e-6))
    qp.set_scalar("proximal", 0)
    qp.vec("dphi", copy=False)[:] = [-v for v in e["ldl_rhs"]]
    L.oq_ldlchol(C.byref(Q), qp.w)
    L.oq_ldlsolveLD_neg_dphi(qp.w)
    assert np.max(np.abs(qp.vec("d") - e["ldl_d"])) <= tol
    qp.set_scalar("proximal", 1)
    qp.set_scalar("gamma", e["ldl_gamma"])
    L.oq_ldlchol(C.byref(Q), qp.w)
    L.oq_ldlsolveLD_neg_dphi(qp.w)
    assert np.max(np.abs(qp.vec("d") - e["ldl_d_prox"])) <= tol


# ---------------------------------------------------------------- suite_lin_alg
def test_lin_alg(golden):
    e = golden["expect"]["lin_alg"]
    L = ob.lib()
    tol = e["tol"]
    fp = ob.fptr

    def abc():
        return ob.f64(e["a"]), ob.f64(e["b"]), np.zeros(3)

    a, b, c = abc()
    L.oq_vec_set_scalar(fp(a), 5.5, 3)
    assert np.all(a == 5.5)
    a, b, c = abc()
    L.oq_vec_self_mult_scalar(fp(a), 3.0, 3)
    assert np.max(np.abs(a - e["self_mult_scalar_3"])) <= tol
    a, b, c = abc()
    for k in range(4):  # the reference's loop never runs (test_lin_alg.c:45, B14); these are its constants
        assert abs(L.oq_vec_prod(fp(a), fp(b), k) - e["vec_prod_expected"][k]) <= tol
    L.oq_vec_add_scaled(fp(a), fp(b), fp(c), 4.0, 3)
    assert np.max(np.abs(c - e["add_scaled_4"])) <= tol
    assert abs(L.oq_vec_norm_inf(fp(a), 3) - e["norm_inf_a"]) <= tol
    assert abs(L.oq_vec_norm_inf(fp(b), 3) - e["norm_inf_b"]) <= tol
    L.oq_vec_ew_recipr(fp(a), fp(c), 3)
    assert np.max(np.abs(c - e["recipr_a"])) <= tol
    L.oq_vec_ew_max_vec(fp(a), fp(b), fp(c), 3)
    assert np.max(np.abs(c - e["max_ab"])) <= tol
    L.oq_vec_ew_min_vec(fp(a), fp(b), fp(c), 3)
    assert np.max(np.abs(c - e["min_ab"])) <= tol
    c[:] = 0
    L.oq_vec_ew_mid_vec(fp(a), fp(c), fp(b), fp(c), 3)
    assert np.max(np.abs(c - e["mid_a_0_b"])) <= tol
    L.oq_vec_ew_prod(fp(a), fp(b), fp(c), 3)
    assert np.max(np.abs(c - e["prod_ab"])) <= tol
    L.oq_vec_ew_div(fp(b), fp(a), fp(c), 3)
    assert np.max(np.abs(c - e["div_ba"])) <= tol
    L.oq_vec_ew_sqrt(fp(b), fp(c), 3)
    assert np.max(np.abs(c - e["sqrt_b"])) <= tol
    # vec_prod grouping (lin_alg.c:72-86): groups of four, then the tail
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(11), rng.standard_normal(11)
    ref = 0.0
    for k in (0, 4):
        ref += (u[k] * v[k] + u[k + 1] * v[k + 1] + u[k + 2] * v[k + 2] + u[k + 3] * v[k + 3])
    for k in range(8, 11):
        ref += u[k] * v[k]
    assert L.oq_vec_prod(fp(u), fp(v), 11) == ref


def test_nonconvex_qp_golden(golden):
    """tests/src/test_nonconvex_qp.c:117-126 on the oracle's restated lobpcg / set_settings_nonconvex"""
    from qpalm_amd.problems import fixture_qp
    e = golden["expect"]["nonconvex_qp"]
    p = fixture_qp(golden["problems"]["nonconvex_qp"])
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(eps_abs=1e-6, eps_rel=1e-6, nonconvex=1, scaling=0, max_rank_update_fraction=1.0, verbose=0))
    o.solve()
    assert o.status_val == 1
    lam = -e["lambda_min"]
    assert abs(o.scalar("gamma") - 1.0 / lam) <= e["gamma_rel_tol"] / lam and 1 / o.scalar("gamma") > lam


def test_blocked_multi_rank_update_is_bit_identical_to_the_rank_one_sweeps():
    """bench.py's cpu_baseline runs the oracle with `updown_block = 8` (up to eight ranks per pass over L, as cholmod_updown carries them:
    oracle/qpalm_oracle.c, dense_ldl_rankk) so that the stated CPU figure is not that of a form the reference's library would not run.
    Every entry receives the operations of the rank-1 routine in the same order: the solve must agree BIT FOR BIT with the scalar form the
    parity tests use -- iterates, multipliers, counts -- on QPs with rank updates of more and of fewer than eight rows."""
    from qpalm_amd.problems import random_qp
    for seed, n, m in ((11, 60, 150), (12, 90, 260)):
        p = random_qp(n, m, seed=seed, density_A=0.08, density_M=0.05)
        res = []
        for block in (0, 8):
            o = ob.OracleQP(*p.args(), settings=ob.default_settings(eps_abs=1e-8, eps_rel=1e-8, verbose=0))
            o.set_scalar("updown_block", block)
            o.solve()
            res.append((o.status_val, int(o.info.iter), o.counter("n_rank1"), o.counter("n_refactor"), o.x.copy(), o.y.copy()))
            o.cleanup()
        assert res[0][:4] == res[1][:4] and res[0][0] == 1 and res[0][2] > 20, (res[0][:4], res[1][:4])
        assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])
