"""Randomised parity campaigns as tests (cases: tests/fuzz_cases.py; the same streams tools/fuzz_parity.py walks).

Campaign G -- random shapes, bound patterns and settings, Schur / KKT / automatic factorisation mixed, on both kernel instances:
status AND iteration count must equal the oracle's, x within 1e-8, y within 1e-8 (1e-5 when sigma_init = 1e3: y <- y + sigma (Ax - z)
multiplies the last bits of Ax by up to sigma_max = 1e9 in BOTH implementations).

Campaign K -- the KKT path forced together with sigma_init = 1e3 (round 2's fuzzing found its divergences only there).  The
quasi-definite matrix then has -1/sigma ~ 1e-9 on its diagonal, the residual that newton.c:64-90 tests against 1e-10 |K sol| -- and,
after an exact Newton step, the inner residual that qpalm.c:515 tests against eps_in -- are rounding noise of the size of the
threshold, and the iteration count is not a property of the algorithm any more: the ORACLE ITSELF changes its count on a few of
these cases when its C source is merely compiled with fused multiply-adds (checked below, in the same test).  The KKT residual is
summed in the reference's order on the device (qpalm_kkt.h); what is left is asserted as: statuses equal on every case; iteration
counts equal on all but at most 2 % of the cases, those within max(4, 5 %) iterations, and the FMA-contracted oracle disagrees
with the plain oracle on cases of the same campaign as well."""
import os
import subprocess

import numpy as np
import pytest

from tests.fuzz_cases import cases, run_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check_general(ctx, seed, count, n_lo, n_hi):
    bad = []
    for it, p, st, warm, meta in cases(seed, count, n_lo, n_hi):
        r = run_case(ctx, p, st, warm)
        ytol = 1e-5 if st["sigma_init"] >= 1e3 else 1e-8
        ok = r["status"][0] == r["status"][1] and r["iter"][0] == r["iter"][1]
        if ok and r["status"][1] in (1, 2):
            ok = r["dx"] <= 1e-8 and r["dy"] <= ytol
        if not ok:
            bad.append((seed, it, meta, {k: st[k] for k in ("factorization_method", "sigma_init", "scaling", "proximal")}, r))
    assert not bad, bad


def test_fuzz_general_campaign(ctx):
    if ctx.kind == "emu":
        _check_general(ctx, 21, 10, 2, 40)
    else:
        for seed in (21, 22, 23, 24):
            _check_general(ctx, seed, 400, 2, 70)                 # the 256-thread instance (factors of at most 256 rows)
        for seed in (31, 32):
            _check_general(ctx, seed, 150, 257, 420)              # the 512-thread instance, KKT panels up to ~1100 rows


def _fma_oracle():
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libqpalm_oracle_fma_%d.so" % os.getpid())
    subprocess.check_call(["gcc", "-O3", "-std=c99", "-fPIC", "-ffp-contract=fast", "-mfma", "-shared", "-o", out,
                           os.path.join(ROOT, "oracle", "qpalm_oracle.c"), "-lm"])
    return out


def test_fuzz_kkt_with_large_sigma(ctx):
    import oracle.binding as ob
    force = dict(factorization_method=0, sigma_init=1e3)
    plan = [(41, 10, 2, 40)] if ctx.kind == "emu" else [(41, 300, 2, 70), (42, 300, 2, 70)]
    fma = _fma_oracle()
    total, iter_off, oracle_unstable = 0, [], 0
    try:
        for seed, count, n_lo, n_hi in plan:
            for it, p, st, warm, meta in cases(seed, count, n_lo, n_hi, force):
                r = run_case(ctx, p, st, warm)
                total += 1
                assert r["status"][0] == r["status"][1], (seed, it, r)
                if r["iter"][0] != r["iter"][1]:
                    assert abs(r["iter"][0] - r["iter"][1]) <= max(4, 0.05 * r["iter"][1]), (seed, it, r)
                    iter_off.append((seed, it, r["iter"]))
                elif r["status"][1] in (1, 2):
                    assert r["dx"] <= 1e-8 and r["dy"] <= 1e-5, (seed, it, r)
                o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st), libpath=fma)
                if warm is not None:
                    o.warm_start(warm[0], warm[1])
                o.solve()
                oracle_unstable += int((o.status_val, int(o.info.iter)) != (r["status"][1], r["iter"][1]))
                o.cleanup()
    finally:
        os.remove(fma)
    assert len(iter_off) <= max(1, int(0.02 * total)), iter_off
    if ctx.kind != "emu":
        # the oracle's own iteration count is not stable under fused multiply-adds on this campaign (4 of 600 cases in round 3)
        assert oracle_unstable >= 1, "the FMA-contracted oracle agreed with the plain one everywhere: tighten this test to exact counts"
