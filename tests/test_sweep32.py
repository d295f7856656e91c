"""The 32-rank multi-pass form of the rank-update sweep (qpalm_dense.h: dense_updown<1, 32>; reference: cholmod_updown behind
ldlupdate_entering_constraints / ldldowndate_leaving_constraints, src/solver_interface.c:407-441).

Rows are handled in passes of one row per thread: pass p first receives the tables of all earlier columns from the export area
("rectangle", no serial chain), then runs the look-ahead sweep on its own triangle.  The 32 ranks of a column are two groups of 16
handled one after the other, i.e. per entry the same operations as two 16-rank sweeps:
  * the factor after an update through the 32-rank form is BIT-IDENTICAL to the one through the 16-rank form (ctx option sweep_ranks),
  * and matches the oracle's entry by entry,
for rank counts around the switch-over points (17, 31, 32, 33, 48, 49, 64, 70), updates, downdates and both at once, sizes with one
pass and with two or three passes (emulator: 128 rows per pass; MI355X: 512), first nonzero rows early and late (skipped passes),
and through a whole solve (fused forward substitution riding on the last sweep: same iterates, bit for bit)."""
import ctypes as C

import numpy as np
import pytest

from oracle import binding as ob
from qpalm_amd.problems import random_qp
from qpalm_amd.solver import QpalmBatch
from tests.test_parity import rel, sizes


def _state(ctx, n, m, seed, sweep_ranks, act):
    ctx.set_option("sweep_ranks", sweep_ranks)
    p = random_qp(n, m, seed=seed, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n))
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    bt.iterate(3)
    bt.set_ivec("active", act)
    bt.op("ldlcholQAtsigmaA")
    return p, st, bt


def _oracle_state(p, st, act):
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**dict(st, max_iter=3)))
    o.solve()
    L = ob.lib()
    ln = ob.c_int(0)
    pa = L.oq_get_ivec(o.w, b"active", C.byref(ln))
    np.ctypeslib.as_array(pa, shape=(len(act),))[:] = act
    L.oq_ldlcholQAtsigmaA(o.w)
    return o, L, pa


CASES = [  # (n_enter, n_leave, rows restricted to those whose first nonzero column is >= this fraction of n)
    (17, 0, 0.0), (32, 0, 0.0), (33, 0, 0.0), (0, 31, 0.0), (0, 49, 0.0), (20, 28, 0.0), (40, 30, 0.0), (64, 0, 0.0), (20, 0, 0.45), (7, 12, 0.0)]


@pytest.mark.parametrize("shape", ["one_pass", "two_pass", "three_pass"])
def test_sweep32_factor_bit_identical_to_sweep16_and_matches_oracle(ctx, shape):
    n, m = {"one_pass": sizes(ctx, (100, 260), (500, 1300)), "two_pass": sizes(ctx, (200, 420), (1000, 2000)),
            "three_pass": sizes(ctx, (250, 500), (1000, 2000))}[shape]
    if shape == "three_pass" and ctx.kind == "hip":
        pytest.skip("the 512-thread instance has at most two passes (1024 rows)")
    rng = np.random.default_rng(99)
    try:
        for (ne, nl, frac) in CASES:
            act = (rng.random(m) < 0.3).astype(np.int64)
            p, st, b32 = _state(ctx, n, m, 4321, 32, act)
            _, _, b16 = _state(ctx, n, m, 4321, 16, act)
            # rows of A by their first nonzero column (the sweep starts at the block of the smallest one)
            first = np.full(m, n)
            for j in range(n):
                rows = p.Ai[p.Ap[j]:p.Ap[j + 1]]
                first[rows] = np.minimum(first[rows], j)
            okrow = first >= frac * n
            enter = np.where((act == 0) & okrow)[0][:ne]
            leave = np.where((act == 1) & okrow)[0][:nl]
            if frac > 0 and len(enter) < 17:
                continue   # too few late rows at this size for a 32-rank sweep
            ne, nl = len(enter), len(leave)
            o, L, pa = _oracle_state(p, st, act)
            ln = ob.c_int(0)
            act_new = act.copy(); act_new[enter] = 1; act_new[leave] = 0
            pao = L.oq_get_ivec(o.w, b"active_old", C.byref(ln))
            np.ctypeslib.as_array(pao, shape=(m,))[:] = act
            np.ctypeslib.as_array(pa, shape=(m,))[:] = act_new
            L.oq_set_entering_leaving_constraints(o.w)
            assert np.array_equal(o.ivec("enter"), enter) and np.array_equal(o.ivec("leave"), leave)
            if ne:
                L.oq_ldlupdate_entering_constraints(o.w)
            if nl:
                L.oq_ldldowndate_leaving_constraints(o.w)
            Lo, Do = o.factor()
            facs = []
            for bt in (b32, b16):
                bt.set_ivec("enter", enter); bt.set_ivec("leave", leave)
                bt.set_scalar("nb_enter", ne); bt.set_scalar("nb_leave", nl)
                if ne:
                    bt.op("ldlupdate_entering_constraints")
                if nl:
                    bt.op("ldldowndate_leaving_constraints")
                facs.append(bt.factor())
            (L32, D32), (L16, D16) = facs
            assert np.array_equal(D32, D16) and np.array_equal(L32, L16), (shape, ne, nl, frac, float(np.max(np.abs(L32 - L16))))
            assert rel(D32, Do) <= 1e-10 and np.max(np.abs(np.tril(L32, -1) - np.tril(Lo, -1))) <= 1e-9, (shape, ne, nl, frac)
            sweeps32, sweeps16 = int(b32.stats(0).n_sweeps), int(b16.stats(0).n_sweeps)
            for bt in (b32, b16):
                bt.close()
            o.cleanup()
    finally:
        ctx.set_option("sweep_ranks", 16)


def test_sweep32_whole_solve_bit_identical(ctx):
    """a whole solve (updates + downdates in one call, fused forward substitution on the last sweep) with 32 and with 16 ranks per
    sweep: identical statuses, counts and iterates bit for bit; fewer sweeps with 32"""
    n, m = sizes(ctx, (200, 420), (1000, 2000))
    probs = [random_qp(n, m, seed=1000 + k, density_A=max(0.01, 4.0 / n), density_M=max(0.005, 2.0 / n)) for k in range(sizes(ctx, 1, 4))]
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    res = {}
    try:
        for sr in (32, 16):
            ctx.set_option("sweep_ranks", sr)
            bt = QpalmBatch(ctx, probs, ctx.default_settings(**st))
            bt.solve()
            res[sr] = (bt.solution(), [(int(i.status_val), int(i.iter)) for i in bt.infos()], [int(s.n_sweeps) for s in bt.stats_all()],
                       [int(s.n_rank1) for s in bt.stats_all()], [int(s.sweep_entries) for s in bt.stats_all()])
            bt.close()
    finally:
        ctx.set_option("sweep_ranks", 16)
    (x32, y32), info32, sw32, r32, e32 = res[32]
    (x16, y16), info16, sw16, r16, e16 = res[16]
    assert info32 == info16 and r32 == r16
    assert np.array_equal(x32, x16) and np.array_equal(y32, y16)
    assert all(a <= b for a, b in zip(sw32, sw16)) and sum(sw32) < sum(sw16), (sw32, sw16)
    for k, p in enumerate(probs[:1]):
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
        o.solve()
        assert info32[k] == (o.status_val, int(o.info.iter)) and rel(x32[k], o.x) <= 1e-9 and rel(y32[k], o.y) <= 1e-9
