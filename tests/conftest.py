import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: the kernels on the host emulator, every GPU thread a fiber) takes twenty minutes on one core and five on six:
    when nobody asked for a process count and pytest-xdist is there, it runs on min(6, cores) workers.  The GPU suite (one device, one context)
    stays in one process; QPALM_TEST_SERIAL=1 keeps the CPU suite there too."""
    try:
        import xdist  # noqa: F401
    except ImportError:
        return None
    opt = config.option
    if (getattr(opt, "markexpr", "") or "").replace(" ", "") != "notgpu" or getattr(opt, "numprocesses", None) is not None or os.environ.get("QPALM_TEST_SERIAL") == "1":
        return None
    if getattr(opt, "collectonly", False) or hasattr(config, "workerinput"):
        return None
    n = min(6, os.cpu_count() or 1)
    if n >= 2:
        opt.numprocesses = n
    return None


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


# ---- counts of the randomised campaigns in the summary (tests/test_fuzz_seeds.py: _report) --------------------------------------------------
_FUZZ_FILE = os.path.join(ROOT, ".pytest_cache", "qpalm_fuzz_counts.jsonl")


def fuzz_count(campaign, backend, total, failed, by_class):
    os.makedirs(os.path.dirname(_FUZZ_FILE), exist_ok=True)
    with open(_FUZZ_FILE, "a") as f:
        f.write(json.dumps(dict(campaign=campaign, backend=backend, total=total, failed=failed, by_class=by_class)) + "\n")


def pytest_sessionstart(session):
    if not hasattr(session.config, "workerinput"):   # the controller (or the only process): a fresh file per run
        try:
            os.remove(_FUZZ_FILE)
        except OSError:
            pass
        if getattr(session.config.option, "numprocesses", None):
            # the test-only libraries are built ONCE, here, before the workers start: several workers finding a stale emulator library used to
            # rebuild it at the same time and load each other's half-written file
            try:
                from qpalm_amd import build
                build.build_oracle()
                build.build_emu()
                build.build_host_emu()
            except Exception as e:   # noqa: BLE001 -- the tests that need a library report the failure themselves
                sys.stderr.write("conftest: pre-building the test libraries failed: %r\n" % (e,))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """One line per randomised campaign that ran: cases, rule failures, and the accepted-but-not-exact cases by bucket (tests/fuzz_cases.py:
    judge_case) -- `rounding` = the reference's own source changes its outcome under the compiler's flags / a one-ulp perturbation / the
    trajectory criterion, `engine-form` = only the oracle variants restating the engine's recurrence move, `singular` = non-finite in the
    engine and in a reference build, `conditioning` = x / y beyond the tolerance but within twice the spread of the reference's own builds.  Everything else
    matched the oracle in status, iteration count, x and y."""
    if not os.path.exists(_FUZZ_FILE):
        return
    rows = [json.loads(ln) for ln in open(_FUZZ_FILE) if ln.strip()]
    if not rows:
        return
    tr = terminalreporter
    tr.write_line("fuzz campaigns (cases / rule failures / accepted as: rounding, engine-form, singular, conditioning):")
    tot = [0, 0, 0, 0, 0, 0]
    for r in rows:
        b = r["by_class"]
        v = [r["total"], r["failed"], b.get("rounding", 0), b.get("engine-form", 0), b.get("singular", 0), b.get("conditioning", 0)]
        tot = [a + c for a, c in zip(tot, v)]
        tr.write_line("  [%s] %-34s %5d / %d / %d, %d, %d, %d" % (r["backend"], r["campaign"], *v))
    tr.write_line("  fuzz total: %d cases, %d rule failures, %d rounding-decided, %d engine-form, %d singular, %d ill-conditioned" % tuple(tot))
