#!/usr/bin/env python3
"""Extracts the golden vectors held by the reference's own test-suite into a JSON fixture.

Run in the build container only (reads /root/reference, which does not exist on the GPU box):

    python tests/golden/extract_reference_fixtures.py

Output: tests/golden/reference_tests.json -- problem data (inputs) and expected outputs
(solution vectors, statuses, tolerances) of tests/src/test_*.c.  Only DATA is extracted: literal
array initialisers and the constants inside mu_assert_* calls, each with its file:line.
"""
import json
import os
import re
import sys

REF = "/root/reference/tests/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_tests.json")

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def read(name):
    with open(os.path.join(REF, name)) as f:
        return f.read()


def define(src, key):
    m = re.search(r"#define\s+%s\s+(\d+)" % key, src)
    return int(m.group(1))


def body_of(src, func_suffix):
    """Text of the first function whose name ends with func_suffix."""
    m = re.search(r"void\s+\w*%s\s*\(void\)\s*\{" % func_suffix, src)
    start = m.end()
    depth, k = 1, start
    while depth:
        ch = src[k]
        depth += ch == "{"
        depth -= ch == "}"
        k += 1
    return src[start:k], src[:start].count("\n") + 1


def assigned(body, lhs, size, cast=float):
    out = [cast(0)] * size
    seen = 0
    for m in re.finditer(r"(?<![\w>])%s\[(\d+)\]\s*=\s*(%s)\s*;" % (re.escape(lhs), NUM), body):
        out[int(m.group(1))] = cast(float(m.group(2)))
        seen += 1
    return out, seen


def brace_list(src, decl):
    m = re.search(re.escape(decl) + r"\s*=\s*\{([^}]*)\}", src)
    return [float(v) for v in re.findall(NUM, m.group(1))]


def problem(fname, data_prefix="data->"):
    src = read(fname)
    N, M = define(src, "N"), define(src, "M")
    body, line0 = body_of(src, "suite_setup")
    anz, qnz = define(src, "ANZMAX"), define(src, "QNZMAX")
    Ap, _ = assigned(body, "Ap", N + 1, int)
    Qp, _ = assigned(body, "Qp", N + 1, int)
    Ai, _ = assigned(body, "Ai", anz, int)
    Ax, _ = assigned(body, "Ax", anz)
    Qi, _ = assigned(body, "Qi", qnz, int)
    Qx, _ = assigned(body, "Qx", qnz)
    if data_prefix is None:  # brace initialisers (test_solver_interface.c:26-30)
        q = brace_list(body, "c_float q[N]")
        bmin = brace_list(body, "c_float bmin[M]")
        bmax = brace_list(body, "c_float bmax[M]")
    else:
        q, _ = assigned(body, data_prefix + "q", N)
        bmin, _ = assigned(body, data_prefix + "bmin", M)
        bmax, _ = assigned(body, data_prefix + "bmax", M)
    return dict(source="tests/src/%s:%d" % (fname, line0), n=N, m=M,
                Ap=Ap, Ai=Ai[:Ap[N]], Ax=Ax[:Ap[N]], Qp=Qp, Qi=Qi[:Qp[N]], Qx=Qx[:Qp[N]],
                q=q, bmin=bmin, bmax=bmax, c=0.0)


def lineno(fname, needle):
    for k, ln in enumerate(read(fname).splitlines(), 1):
        if needle in ln:
            return "tests/src/%s:%d" % (fname, k)
    raise KeyError(needle)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; the committed JSON is the fixture")
    fx = {"_generator": "tests/golden/extract_reference_fixtures.py", "problems": {}, "expect": {}}
    P = fx["problems"]
    P["basic_qp"] = problem("test_basic_qp.c")
    P["medium_qp"] = problem("test_medium_qp.c")
    P["degen_hess"] = problem("test_degen_hess.c")
    P["ls_qp"] = problem("test_ls_qp.c")
    P["prim_inf_qp"] = problem("test_prim_inf_qp.c")
    P["dua_inf_qp"] = problem("test_dua_inf_qp.c")
    P["update"] = problem("test_update.c")
    P["nonconvex_qp"] = problem("test_nonconvex_qp.c")
    P["error_handling"] = problem("test_error_handling.c")
    P["solver_interface"] = problem("test_solver_interface.c", data_prefix=None)

    E = fx["expect"]
    E["basic_qp"] = dict(
        solution=brace_list(read("test_basic_qp.c"), "static c_float solution[N]"),
        rel_tol=1e-5, source=lineno("test_basic_qp.c", "static c_float solution[N]"),
        settings=dict(eps_abs=1e-6, eps_rel=1e-6, gamma_init=1e1, max_rank_update_fraction=1.0),
        settings_source=lineno("test_basic_qp.c", "settings->gamma_init = 1e1"),
        warm_x=[2.0, -60.0, -3380.0, -6.0],
        warm_y_scaled=[0.0, 0.0, -23.0, -0.014, 0.0], warm_y=[0.0, 0.0, -23.0, -0.01, 0.0],
        warm_source=lineno("test_basic_qp.c", "c_float x[N] = {2.0, -60.0, -3380.0, -6.0};"),
        warm_iter_lt=12, resolve_tol=1e-15,
        dual_objective_tol=1e-5, dual_limit_early=-1000000000.0, sigma_max_test=1e3, time_limit_test=0.01 * 1e-3)
    E["medium_qp"] = dict(
        solution=brace_list(read("test_medium_qp.c"), "static c_float solution[N]"), rel_tol=1e-5,
        source=lineno("test_medium_qp.c", "static c_float solution[N]"),
        settings=dict(eps_abs=1e-6, eps_rel=1e-6, max_rank_update_fraction=1.0))
    E["degen_hess"] = dict(solution=[5.5, 5.0, -10.0], abs_tol=1e-5,
                           source=lineno("test_degen_hess.c", "mu_assert_double_eq(work->solution->x[0], 5.5, TOL)"),
                           settings=dict(eps_abs=1e-6, eps_rel=1e-6, max_rank_update_fraction=1.0))
    E["ls_qp"] = dict(solution=[-2.0, -2.0e4], abs_tol=1e-5,
                      source=lineno("test_ls_qp.c", "solution[1] = -2.0000000e+04"),
                      settings=dict(eps_abs=1e-6, eps_rel=1e-6, gamma_max=1e3, gamma_init=1e1, max_rank_update_fraction=1.0))
    E["prim_inf_qp"] = dict(status="PRIMAL_INFEASIBLE", source=lineno("test_prim_inf_qp.c", "QPALM_PRIMAL_INFEASIBLE"),
                            settings=dict(eps_abs=1e-6, eps_rel=1e-6, max_rank_update_fraction=1.0),
                            variants=[dict(proximal=1, scaling=2), dict(proximal=1, scaling=0),
                                      dict(proximal=0, scaling=2), dict(proximal=0, scaling=0)])
    E["dua_inf_qp"] = dict(status="DUAL_INFEASIBLE", source=lineno("test_dua_inf_qp.c", "QPALM_DUAL_INFEASIBLE"),
                           settings=dict(eps_abs=1e-6, eps_rel=1e-6, max_rank_update_fraction=1.0),
                           variants=[dict(proximal=1, scaling=2), dict(proximal=1, scaling=0),
                                     dict(proximal=0, scaling=2), dict(proximal=0, scaling=0)])
    E["update"] = dict(settings=dict(eps_abs=1e-6, eps_rel=1e-6, scaling=2, proximal=1),
                       first=[-0.1, 0.3], after_bounds=[0.0, 0.15], after_q=[0.02, 0.18], abs_tol=1e-5,
                       new_bmin0=0.0, new_bmax1=1.5, new_q=[-0.5, -0.75],
                       update_settings=dict(gamma_init_factor=0.1, theta=0.9, proximal=1, scaling=10),
                       source=lineno("test_update.c", "mu_assert_double_eq(work->solution->x[0], -0.1, 1e-5)"))
    E["nonconvex_qp"] = dict(lambda_min=-0.0021544347, gamma_rel_tol=1e-1,
                             source=lineno("test_nonconvex_qp.c", "mu_assert_double_eq(work->gamma"))
    E["solver_interface"] = dict(
        x=[1.1, -0.5], Ad_in=[1.1, -0.5, 20.0], A_x=[0.1, 1.3, 5.5], Q_x=[1.6, -2.1], At_Ad=[99.6, 0.2],
        inf_norm_cols=[5.0, 4.0], inf_norm_rows=[2.0, 4.0, 5.0],
        ldl_rhs=[1.0, 2.0], ldl_d=[4.0, 3.0], ldl_gamma=1e3,
        ldl_d_prox=[3.989028924198480, 2.993017953122679], tol=1e-8,
        source=lineno("test_solver_interface.c", "mu_assert_double_eq(work->Ad[0], 0.1, TOL)"))
    E["lin_alg"] = dict(
        a=[0.1, 2.5, -3.9], b=[0.0, 10.0, 4.0], tol=1e-8,
        self_mult_scalar_3=[0.3, 7.5, -11.7], add_scaled_4=[0.1, 42.5, 12.1], norm_inf_a=3.9, norm_inf_b=10.0,
        recipr_a=[10.0, 0.4, -0.256410256410256], max_ab=[0.1, 10.0, 4.0], min_ab=[0.0, 2.5, -3.9],
        mid_a_0_b=[0.0, 2.5, 0.0], prod_ab=[0.0, 25.0, -15.6], div_ba=[0.0, 4.0, -1.025641025641026],
        sqrt_b=[0.0, 3.162277660168380, 2.0], vec_prod_expected=[0.0, 0.0, 25.0, 9.4],
        source=lineno("test_lin_alg.c", "a[0] = 0.1; a[1] = 2.5; a[2] = -3.9;"))
    with open(OUT, "w") as f:
        json.dump(fx, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
