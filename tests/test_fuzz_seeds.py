"""Randomised parity campaigns as tests (cases: tests/fuzz_cases.py; the same streams tools/fuzz_parity.py walks).

The assertion, for EVERY case of every campaign (tests/fuzz_cases.py: judge_case):
  * status AND iteration count equal the oracle's, x within 1e-8, y within max(1e-8, 100 sigma dx) where sigma is the largest penalty the
    solve reached and dx the difference of x: y <- y + sigma (Ax - z) multiplies the last bits of x by the penalty in BOTH
    implementations, the multipliers may differ by what the difference of x explains and by nothing more; or
  * the case is one whose count rounding decides: the ORACLE ITSELF, its C source compiled with fused multiply-adds (-ffp-contract=fast
    -mfma) or -Ofast, does not reproduce its own (status, iterations) -- then the engine's status must be one an oracle variant reaches
    and, when both solved, the objectives agree to 10 x the case's tolerance.
Round 3 allowed "at most 2 % of the KKT cases within max(4, 5 %) iterations" without comparing anything else on them; that is gone.

Campaign G -- random shapes, bound patterns and settings, Schur / KKT / automatic factorisation mixed, on both kernel instances.
Campaign K -- the KKT path forced together with sigma_init = 1e3 (the quasi-definite matrix then has -1/sigma ~ 1e-9 on its diagonal
and the inner termination test compares rounding noise with its threshold: most rounding-decided cases live here).
Named cases -- the five mismatches of round 3's end-of-round campaign (profiles/r03/fuzz), pinned."""
import numpy as np
import pytest

from tests.fuzz_cases import cases, judge_case, run_case


def _campaign(ctx, seed, count, n_lo, n_hi, force=None):
    bad, soft = [], []
    for it, p, st, warm, meta in cases(seed, count, n_lo, n_hi, force):
        r = run_case(ctx, p, st, warm)
        ok, why, rounding = judge_case(r, p, st, warm, 1e-8, ctx)
        if not ok:
            bad.append((seed, it, meta, {k: st[k] for k in ("factorization_method", "sigma_init", "scaling", "proximal")}, why))
        elif rounding:
            soft.append((seed, it, why))
    return bad, soft


def test_fuzz_general_campaign(ctx):
    plan = [(21, 10, 2, 40)] if ctx.kind == "emu" else [(s, 400, 2, 70) for s in (21, 22, 23, 24)] + [(s, 150, 257, 420) for s in (31, 32)]
    bad, soft, total = [], [], 0
    for seed, count, n_lo, n_hi in plan:   # 2..70: the 256-thread instance; 257..420: the 512-thread instance, KKT panels up to ~1100 rows
        b, s = _campaign(ctx, seed, count, n_lo, n_hi)
        bad += b; soft += s; total += count
    assert not bad, bad
    assert len(soft) <= max(1, total // 100), soft   # rounding-decided cases are rare (round 3: 5 of 1960 fresh cases)


def test_fuzz_kkt_with_large_sigma(ctx):
    force = dict(factorization_method=0, sigma_init=1e3)
    plan = [(41, 10, 2, 40)] if ctx.kind == "emu" else [(41, 300, 2, 70), (42, 300, 2, 70)]
    bad, soft, total = [], [], 0
    for seed, count, n_lo, n_hi in plan:
        b, s = _campaign(ctx, seed, count, n_lo, n_hi, force)
        bad += b; soft += s; total += count
    assert not bad, bad
    assert len(soft) <= max(1, total // 40), soft


def test_sigma_grown_by_one_ulp(ctx):
    """Seed 204 case 155 of round 4's fresh-seed campaign: one sigma_k grows by one unit in the last place, sqrt(mult_factor) == 1, and the
    reference's CHOLMOD branch of ldlupdate_sigma_changed would scale the zeroed row of At_sqrt_sigma back by 1/0 (solver_interface.c:
    492-502).  The engine returned NaN (MAX_ITER) there through round 3; engine and oracle now rebuild the row: SOLVED in 41 iterations."""
    for it, p, st, warm, meta in cases(204, 156, 2, 70):
        if it != 155:
            continue
        r = run_case(ctx, p, st, warm)
        assert r["status"] == (1, 1) and r["iter"][0] == r["iter"][1] and r["dx"] <= 1e-8 and r["dy"] <= 1e-8, r


# (seed, case, n_lo, n_hi) of tests/fuzz_cases.py: the five cases of round 3's fresh-seed campaign on which engine and oracle differed
R03_MISMATCHES = [(101, 314, 2, 70), (101, 448, 2, 70), (103, 466, 2, 70), (111, 48, 257, 600), (112, 158, 257, 600)]


@pytest.mark.parametrize("seed,case,n_lo,n_hi", R03_MISMATCHES)
def test_round3_mismatches_are_rounding_decided(ctx, seed, case, n_lo, n_hi):
    if ctx.kind == "emu" and n_lo > 100:
        pytest.skip("n > 400: minutes in the emulator; runs on the hardware")
    for it, p, st, warm, meta in cases(seed, case + 1, n_lo, n_hi):
        if it != case:
            continue
        r = run_case(ctx, p, st, warm)
        ok, why, rounding = judge_case(r, p, st, warm, 1e-8, ctx)
        assert ok, (seed, case, meta, why, r)
