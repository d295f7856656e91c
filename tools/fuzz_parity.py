"""Randomised parity fuzzing (cases: tests/fuzz_cases.py), the engine against the oracle: status and iteration counts exact, x and
y to 1e-8.  TEST TOOL (uses oracle/): python tools/fuzz_parity.py <seed> <cases> [hip|emu] [n_lo n_hi] [key=value ...]
(key=value pairs force settings, e.g. factorization_method=0 sigma_init=1e3)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qpalm_amd.solver import Context  # noqa: E402
from tests.fuzz_cases import cases, run_case  # noqa: E402

pos = [a for a in sys.argv[1:] if "=" not in a]
force = {}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        force[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
backend = pos[2] if len(pos) > 2 else 'hip'
ctx = Context(0, lib_path=os.path.join(ROOT, 'tests', 'emu', 'libqpalm_gfx950_emu.so')) if backend == 'emu' else Context(0, lib_path=os.environ.get('QPALM_LIB') or None)
seed = int(pos[0]) if len(pos) > 0 else 0
N = int(pos[1]) if len(pos) > 1 else 100
NLO, NHI = (int(pos[3]), int(pos[4])) if len(pos) > 4 else (2, 70)   # range of n (m up to 1.7 n)
bad = 0    # status or iteration count differs (or an exception)
soft = 0   # same status and iterations, x or y beyond 1e-8
t0 = time.time()
for it, p, st, warm, meta in cases(seed, N, NLO, NHI, force or None):
    try:
        r = run_case(ctx, p, st, warm)
        hard_ok = r["status"][0] == r["status"][1] and r["iter"][0] == r["iter"][1]
        ok = hard_ok
        if ok and r["status"][1] in (1, 2):
            ok = r["dx"] <= 1e-8 and r["dy"] <= 1e-8
        if not ok:
            if hard_ok:
                soft += 1
            else:
                bad += 1
            print("MISMATCH" if not hard_ok else "SOFT", "seed", seed, "case", it, meta, st, "status", r["status"], "iter", r["iter"], "x", r["dx"], "y", r["dy"])
            sys.stdout.flush()
    except Exception as e:
        bad += 1
        print("EXC", it, meta, st, repr(e)[:300])
print("done seed", seed, "cases", N, "status/iteration mismatches", bad, "x/y beyond 1e-8", soft, "time", round(time.time() - t0, 1))
