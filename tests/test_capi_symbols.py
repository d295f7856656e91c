"""CPU checks of the shipped C-ABI library: it loads, exports every symbol include/qpalm_gfx950.h
declares, and FAILS LOUDLY without a GPU (no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import pytest

from qpalm_amd import build, capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    return capi.load(build.build_hip())


def test_header_symbols_are_exported(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "qpalm_gfx950.h")).read()
    declared = set(re.findall(r"\b(qpg_[a-zA-Z_0-9]+)\s*\(", hdr))
    declared -= {"qpg_ctx", "qpg_batch"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in declared:
        assert hasattr(hip_lib, s), s
    assert hip_lib.qpg_backend_name() == b"gfx950-hip"


def test_settings_defaults_and_validation(hip_lib):
    s = capi.Settings()
    hip_lib.qpg_set_default_settings(C.byref(s))
    # include/constants.h:65-110
    assert (s.max_iter, s.inner_max_iter, s.scaling, s.max_rank_update) == (10000, 100, 10, 160)
    assert (s.eps_abs, s.eps_rel, s.rho, s.theta, s.delta) == (1e-4, 1e-4, 0.1, 0.25, 100)
    assert (s.sigma_max, s.sigma_init, s.gamma_init, s.gamma_upd, s.gamma_max) == (1e9, 2e1, 1e7, 10, 1e7)
    assert hip_lib.qpg_validate_settings(C.byref(s)) == 1
    s.max_iter = -1
    assert hip_lib.qpg_validate_settings(C.byref(s)) == 0


def test_no_gpu_fails_loudly(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = hip_lib.qpg_ctx_create(0, C.byref(h))
    assert rc == -1  # QPG_ERR_NO_DEVICE
    assert b"no CPU fallback" in hip_lib.qpg_last_error()


def test_product_loader_has_no_fallback(tmp_path):
    with pytest.raises(ImportError):
        capi.load(str(tmp_path / "missing.so"))


def test_large_factors_get_the_lds_their_solve_needs(hip_lib):
    """qpg_batch_create is host logic (no device memory before setup): dense_solve keeps the right-hand side behind its two 32 x 32
    tiles in LDS, so a batch whose factor has close to 8192 rows (n + m in KKT mode) must ask for more than the default block, and
    more than 8192 rows are refused."""
    s = capi.Settings()
    hip_lib.qpg_set_default_settings(C.byref(s))
    s.factorization_method = 0          # FACTORIZE_KKT: factor rows = n + m
    ctx = C.c_void_p(1)                 # never dereferenced before a device call... but create needs a real context:
    # a context cannot exist without a GPU, so drive the same code through the emulated build of the same source
    from qpalm_amd.solver import Context
    ectx = Context(0, lib_path=build.build_emu())
    L = ectx.L
    for n, m, ok in ((4000, 4190, True), (1000, 2000, True), (4096, 4097, False)):
        h = C.c_void_p()
        rc = L.qpg_batch_create(ectx.h, 1, n, m, 10, 10, C.byref(s), C.byref(h))
        assert (rc == 0) == ok, (n, m, rc)
        if ok:
            w, t, l = capi.c_int(0), capi.c_int(0), capi.c_int(0)
            assert L.qpg_batch_launch_shape(h, C.byref(w), C.byref(t), C.byref(l)) == 0
            assert l.value >= 2 * 32 * 33 * 8 + 256 + 8 * (n + m), (n, m, l.value)
            L.qpg_batch_destroy(h)
