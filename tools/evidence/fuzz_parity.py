"""Randomised parity fuzzing (cases: tests/fuzz_cases.py), the engine against the oracle, judged by the rule of tests/test_fuzz_seeds.py
(judge_case: status and iteration count exact, x to 1e-8, y to max(1e-8, 100 sigma dx) -- unless the oracle's own outcome
depends on its floating-point contraction, then status among the variants' and objectives to 10 x eps).  TEST TOOL (uses oracle/): python tools/evidence/fuzz_parity.py <seed> <cases> [hip|emu] [n_lo n_hi] [key=value ...]
(key=value pairs force settings, e.g. factorization_method=0 sigma_init=1e3; sparse=1: the engine's sparse factor against the oracle's sparse-storage mode,
ordering=1 with it: under the engine's nested-dissection ordering)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qpalm_amd.solver import Context  # noqa: E402
from tests.fuzz_cases import cases, judge_case, run_case  # noqa: E402

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
sparse = int(force.pop("sparse", 0))
for opt in ("small_workgroups", "linesearch_hbm", "coop", "coop_rank_threshold", "coop_updates", "sequential_rank_sums"):   # engine options, not settings (the 128-thread instance; the tiled line-search sort with tiles of that many entries; coop mode and its one-launch sweep)
    if opt in force:
        ctx.set_option(opt, int(force.pop(opt)))
ordering = force.pop("ordering", None)   # with sparse=1: 0 natural, 1 nested dissection (default: the library's automatic choice)
if sparse:
    ctx.set_option("sparse_factor", 1)
    if ordering is not None:
        ctx.set_option("sparse_ordering", int(ordering))
    force.update(dict(factorization_method=1))
worst_dy_ratio = 0.0   # largest dy / derived bound among the cases that needed more than 1e-8 on y
bad = 0    # fails the rule (or an exception)
soft = 0   # passes, but not as an outright match: by bucket in `by` (rounding / engine-form / singular: tests/fuzz_cases.py, judge_case)
by = {}
guards = [0, 0]   # Newton steps redone with a fresh factorisation (engine, oracle)
t0 = time.time()
for it, p, st, warm, meta in cases(seed, N, NLO, NHI, force or None):
    try:
        r = run_case(ctx, p, st, warm, oracle_sparse_mode=1 if sparse else 0)
        ok, why, rounding = judge_case(r, p, st, warm, 1e-8, ctx if not sparse else None)
        if r.get("ytol_used", 0) > 1e-8 and r["dy"] > 1e-8:
            worst_dy_ratio = max(worst_dy_ratio, r["dy"] / r["ytol_used"])
        guards[0] += r["guard"][0]; guards[1] += r["guard"][1]
        if not ok or rounding:
            bad += 0 if ok else 1
            soft += 1 if ok else 0
            if ok:
                by[rounding if isinstance(rounding, str) else "rounding"] = by.get(rounding if isinstance(rounding, str) else "rounding", 0) + 1
            print("FAIL" if not ok else {"rounding": "ROUNDING-DECIDED", "engine-form": "ENGINE-FORM", "singular": "SINGULAR", "conditioning": "ILL-CONDITIONED"}.get(rounding, "ROUNDING-DECIDED"), "seed", seed, "case", it, meta, {k: st[k] for k in ("factorization_method", "sigma_init", "scaling", "proximal")},
                  "status", r["status"], "iter", r["iter"], "x", r["dx"], "y", r["dy"], "|", why)
            sys.stdout.flush()
    except Exception as e:
        bad += 1
        print("EXC", it, meta, st, repr(e)[:300])
print("done seed", seed, "cases", N, "n", (NLO, NHI), "forced", force, "FAIL", bad, "accepted-not-exact", soft, by, "newton steps redone (engine, oracle)", guards, "largest dy / derived y bound", round(worst_dy_ratio, 3), "time", round(time.time() - t0, 1))
