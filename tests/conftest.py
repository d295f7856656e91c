import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Same order as bench.py: torch initialises the HIP runtime first, the solver library joins it.
    # (The other way round torch reports "No HIP GPUs are available" in this image.)  No-op without a GPU.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_tests.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY host build of the kernel source (tests/emu); never used by the product."""
    from qpalm_amd import build
    return build.build_emu()


_CTX = {}


@pytest.fixture(params=[pytest.param("emu", id="emu"), pytest.param("hip", marks=pytest.mark.gpu, id="hip")])
def ctx(request):
    """A backend context: 'emu' = kernel source emulated on the host (CPU tests),
    'hip' = the shipped gfx950 library on a real MI355X (the parity tests proper)."""
    from qpalm_amd.solver import Context
    kind = request.param
    if kind not in _CTX:
        if kind == "emu":
            _CTX[kind] = Context(0, lib_path=request.getfixturevalue("emu_lib"))
            assert _CTX[kind].backend == "host-emulation"
        else:
            _CTX[kind] = Context(0)
            assert _CTX[kind].backend == "gfx950-hip"
    c = _CTX[kind]
    c.kind = kind
    c.set_option("update_rank_threshold", -1)
    c.set_option("max_slots", 512)
    c.set_option("place_panel_wave", 0)
    c.set_option("sweep_ranks", 16)
    c.set_option("kkt_compact", 1)
    c.set_option("coop", 0)              # the one-workgroup-per-QP engine; tests/test_coop.py switches the multi-workgroup mode on
    c.set_option("coop_workgroups", 3 if kind == "emu" else 256)
    return c
