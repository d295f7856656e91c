"""Builds and runs tests/c/test_host_api.c: the reference's C suites against include/qpalm_host.h
(qpalm_setup / qpalm_solve / QPALMWorkspace through the C host library qpalm_amd/host/qpalm_host.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CDIR = os.path.join(ROOT, "tests", "c")


def _emit_header(golden):
    out = ["/* generated from tests/golden/reference_tests.json by tests/test_host_c_api.py */", "#include <stdint.h>",
           "typedef struct { size_t n, m, nnzA, nnzQ; const int64_t *Ap, *Ai, *Qp, *Qi; const double *Ax, *Qx, *q, *bmin, *bmax; } golden_problem;"]

    def arr(t, name, v):
        return "static const %s %s[] = {%s};" % (t, name, ", ".join(repr(x) for x in (v if len(v) else [0])))
    for name in ("basic_qp", "degen_hess", "prim_inf_qp", "dua_inf_qp", "update", "solver_interface", "nonconvex_qp"):
        p = golden["problems"][name]
        for k in ("Ap", "Ai", "Qp", "Qi"):
            out.append(arr("int64_t", "%s_%s" % (name, k), [int(x) for x in p[k]]))
        for k in ("Ax", "Qx", "q", "bmin", "bmax"):
            out.append(arr("double", "%s_%s" % (name, k), [float(x) for x in p[k]]))
        out.append("static const golden_problem golden_%s = {%d, %d, %d, %d, %s_Ap, %s_Ai, %s_Qp, %s_Qi, %s_Ax, %s_Qx, %s_q, %s_bmin, %s_bmax};"
                   % ((name, p["n"], p["m"], len(p["Ax"]), len(p["Qx"])) + (name,) * 9))
    out.append(arr("double", "basic_qp_solution", golden["expect"]["basic_qp"]["solution"]))
    with open(os.path.join(CDIR, "golden_data.h"), "w") as f:
        f.write("\n".join(out) + "\n")


def _build_and_run(libdir, backend_lib, host_out):
    src = os.path.join(ROOT, "qpalm_amd", "host", "qpalm_host.c")
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-shared", "-Wall", "-o", host_out, src,
                           os.path.join(ROOT, "qpalm_amd", "host", "qpalm_qps.c"), "-L" + libdir, "-l" + backend_lib, "-Wl,-rpath," + libdir, "-lm"])
    exe = os.path.join(CDIR, "test_host_api_" + backend_lib)
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-Wall", "-o", exe, os.path.join(CDIR, "test_host_api.c"), host_out,
                           "-Wl,-rpath," + os.path.dirname(host_out), "-Wl,-rpath," + libdir, "-L" + libdir, "-l" + backend_lib, "-lm"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0, r.stdout[-3000:]
    assert "0 failures" in r.stdout


def test_reference_c_suites_on_emulated_kernels(golden, emu_lib):
    _emit_header(golden)
    d = os.path.dirname(emu_lib)
    _build_and_run(d, "qpalm_gfx950_emu", os.path.join(d, "libqpalm_host_emu.so"))


@pytest.mark.gpu
def test_reference_c_suites_on_gfx950(golden):
    from qpalm_amd import build
    _emit_header(golden)
    d = os.path.dirname(build.LIB)
    _build_and_run(d, "qpalm_gfx950", os.path.join(d, "libqpalm.so"))
