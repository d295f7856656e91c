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
