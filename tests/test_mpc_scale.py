"""BASELINE.json configs[2] at config scale: a batch of mpc-160 QPs (n=160, m=270; T=10, nx=10, nu=5) driven through a
receding-horizon sequence as simulations/randomMPCsequential.m:158-177 does: apply the first input, shift the previous
solution as warm start, move the initial-state bounds (qpalm_update_bounds) and solve again.  Every step a sample of
the batch is compared with the oracle driven through the same calls (iterations and active sets exact, x, y to 1e-9);
all QPs must be solved and feasible.

[emu]: 6 QPs x 3 steps on the host-emulated kernels (CPU suite).  [hip]: 4096 QPs x 5 steps, 64-QP oracle sample."""
import numpy as np
import pytest

from oracle import binding as ob
from qpalm_amd.problems import random_mpc_qp
from qpalm_amd.solver import QpalmBatch
from tests.test_parity import rel, sizes, RTOL

ST = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
T, NX, NU = 10, 10, 5


def _plants(nb, per_plant):
    """one random plant per `per_plant` QPs; returns (problems, dynamics) with every QP its own initial state"""
    rng = np.random.default_rng(2024)
    probs, dyn = [], []
    base = None
    for k in range(nb):
        if k % per_plant == 0:
            base = random_mpc_qp(T=T, nx=NX, nu=NU, seed=100 + k // per_plant)
            A = base.A_mat().toarray()
            # rows nx..2nx-1 hold  x_1 - Adyn x_0 - Bdyn u_0 = 0
            Adyn = -A[NX:2 * NX, 0:NX]
            Bdyn = -A[NX:2 * NX, (T + 1) * NX:(T + 1) * NX + NU]
        x0 = 2.0 * (2 * rng.random(NX) - 1)
        bmin, bmax = base.bmin.copy(), base.bmax.copy()
        bmin[:NX] = x0
        bmax[:NX] = x0
        probs.append(type(base)(base.n, base.m, base.Qp, base.Qi, base.Qx, base.Ap, base.Ai, base.Ax, base.q, bmin, bmax))
        dyn.append((Adyn, Bdyn))
    return probs, dyn, rng


def _shift(x, Adyn):
    """randomMPCsequential.m:166-168 in this QP's variable order [x_0..x_T, u_0..u_{T-1}]"""
    xs, us = x[:(T + 1) * NX].reshape(T + 1, NX), x[(T + 1) * NX:].reshape(T, NU)
    xs2 = np.vstack([xs[1:], (Adyn @ xs[-1])[None, :]])
    us2 = np.vstack([us[1:], np.zeros((1, NU))])
    return np.concatenate([xs2.ravel(), us2.ravel()])


@pytest.mark.parametrize("mode", ["schur", "kkt"])
def test_mpc_sequence_at_config_scale(ctx, mode):
    """mode kkt: FACTORIZE_KKT, i.e. the (n+m) x (n+m) panel with row additions / deletions -- what BASELINE.json's config 3 names
    literally ("warm-start row add/delete LDL' updates"); [hip]: 1024 QPs x 4 steps there, a 32-QP oracle sample."""
    kkt = mode == "kkt"
    nb, nsteps, nsample, per_plant = sizes(ctx, (1, 2, 1, 1), (1024, 4, 32, 64)) if kkt else sizes(ctx, (3, 2, 3, 3), (4096, 5, 64, 64))
    ST = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, **(dict(factorization_method=0) if kkt else {}))
    probs, dyn, rng = _plants(nb, per_plant)
    n, m = probs[0].n, probs[0].m
    assert (n, m) == (160, 270)
    sample = np.unique(np.linspace(0, nb - 1, nsample).astype(int))
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**ST))
    oracles = {k: ob.OracleQP(*probs[k].args(), settings=ob.default_settings(**ST)) for k in sample}
    bmin = np.stack([p.bmin for p in probs])
    bmax = np.stack([p.bmax for p in probs])
    bt.solve()
    for o in oracles.values():
        o.solve()
    for step in range(nsteps):
        xs, ys = bt.solution()
        infos = bt.infos()
        assert all(int(i.status_val) == 1 for i in infos), "step %d: not every QP solved" % step
        for k in sample:
            o = oracles[k]
            assert o.status_val == 1
            assert int(infos[k].iter) == int(o.info.iter) and int(infos[k].iter_out) == int(o.info.iter_out), (step, k)
            tol = 1e-8 if kkt else RTOL   # the quasi-definite KKT systems of warm-started steps (-1/sigma down to 1e-9 on the diagonal) lose a digit
            assert rel(xs[k], o.x) <= tol and rel(ys[k], o.y) <= tol, (step, k, rel(xs[k], o.x), rel(ys[k], o.y))
            assert np.array_equal(bt.ivec("active", k), o.ivec("active")), (step, k)
        # size-independent property on the whole batch: the dynamics rows hold (A x = 0 for rows nx..)
        if step == nsteps - 1:
            break
        # the plant moves: apply u_0 (+ disturbance), shift the solution, move the x_0 bounds, warm start
        xw = np.empty_like(xs)
        for k in range(nb):
            Adyn, Bdyn = dyn[k]
            u0 = xs[k, (T + 1) * NX:(T + 1) * NX + NU]
            x_init = Adyn @ xs[k, :NX] + Bdyn @ u0 + 1e-2 * rng.standard_normal(NX)
            bmin[k, :NX] = x_init
            bmax[k, :NX] = x_init
            xw[k] = _shift(xs[k], Adyn)
        assert bt.update_bounds(bmin, bmax) == 0
        bt.warm_start(xw, ys)
        bt.solve()
        for k in sample:
            o = oracles[k]
            o.update_bounds(bmin[k], bmax[k])
            o.warm_start(xw[k], ys[k])
            o.solve()
    for o in oracles.values():
        o.cleanup()


def test_warm_start_last_equals_host_round_trip(ctx):
    """qpg_batch_warm_start_last (previous solution straight from HBM) == qpalm_warm_start(solution()) bit for bit, and a
    second qpg_batch_solve on finished QPs starts over (re-armed on the device, src/qpalm.c:401-420)."""
    nb = sizes(ctx, 1, 64)
    probs, dyn, rng = _plants(nb, per_plant=max(1, nb // 4))
    st = ctx.default_settings(**ST)
    a, b = QpalmBatch(ctx, probs, st), QpalmBatch(ctx, probs, st)
    for bt in (a, b):
        bt.solve()
    xa, ya = a.solution()
    it0 = [int(i.iter) for i in a.infos()]
    bmin = np.stack([p.bmin for p in probs]); bmax = np.stack([p.bmax for p in probs])
    bmin[:, :NX] += 0.05; bmax[:, :NX] += 0.05
    for bt in (a, b):
        bt.update_bounds(bmin, bmax)
    a.warm_start(xa, ya)
    b.warm_start_last()
    for bt in (a, b):
        bt.solve()
    (x1, y1), (x2, y2) = a.solution(), b.solution()
    assert np.array_equal(x1, x2) and np.array_equal(y1, y2)
    assert [int(i.iter) for i in a.infos()] == [int(i.iter) for i in b.infos()]
    assert all(int(i.status_val) == 1 for i in b.infos())
    # solve again without a warm start call: every finished QP starts a new solve from its current iterate
    b.solve()
    assert all(int(i.status_val) == 1 for i in b.infos()) and all(int(i.iter) >= 1 for i in b.infos())
    assert len(it0) == nb


def test_pinned_host_arrays(ctx):
    """qpg_host_alloc: bounds handed over from, and solutions taken into, page-locked arrays give the same results."""
    nb = sizes(ctx, 1, 32)
    probs, dyn, rng = _plants(nb, per_plant=max(1, nb // 4))
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**ST))
    bt.solve()
    x, y = bt.solution()
    px, py = ctx.pinned_array((nb, probs[0].n)), ctx.pinned_array((nb, probs[0].m))
    assert px.shape == x.shape and not px.any()
    rx, ry = bt.solution(out=(px, py))
    assert rx is px and np.array_equal(px, x) and np.array_equal(py, y)
    bmin, bmax = ctx.pinned_array((nb, probs[0].m)), ctx.pinned_array((nb, probs[0].m))
    bmin[:] = np.stack([p.bmin for p in probs]); bmax[:] = np.stack([p.bmax for p in probs])
    bmin[:, :NX] += 0.02; bmax[:, :NX] += 0.02
    other = QpalmBatch(ctx, probs, ctx.default_settings(**ST))
    other.solve()
    for b in (bt, other):
        b.warm_start_last()
    bt.update_bounds(bmin, bmax)
    other.update_bounds(np.array(bmin), np.array(bmax))
    bt.solve(); other.solve()
    assert np.array_equal(bt.solution()[0], other.solution()[0])
