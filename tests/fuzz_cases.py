"""Randomised parity cases (TEST INFRASTRUCTURE, used by tools/fuzz_parity.py and tests/test_fuzz_seeds.py): small random QPs with
random shapes, bound patterns, settings (scaling, proximal, sigma, gamma, dual termination, KKT / Schur, inner_max_iter,
max_iter) and warm starts.  `cases(seed, count, n_lo, n_hi)` yields (index, problem, settings, warm_start) in a fixed order:
case k of a seed is always the same problem (the draws of a case do not depend on any solve)."""
import numpy as np

from qpalm_amd.problems import random_qp


def cases(seed, count, n_lo=2, n_hi=70, force=None):
    """force: dict of settings that overrides the drawn ones (the draws are made all the same, so the stream stays aligned)"""
    rng = np.random.default_rng(int(seed))
    for it in range(count):
        n = int(rng.integers(n_lo, n_hi))
        m = int(rng.integers(1, max(2, int(1.7 * n_hi))))
        dA = float(rng.choice([0.05, 0.15, 0.4, 1.0])) * min(1.0, 70.0 / n)
        dM = float(rng.choice([0.02, 0.1, 0.5])) * min(1.0, 70.0 / n)
        p = random_qp(n, m, seed=int(rng.integers(1 << 30)), density_A=dA, density_M=dM)
        mode = int(rng.integers(0, 4))          # widen / tighten / equality / infinite bounds
        if mode == 1:
            p.bmax[:] = p.bmin + 0.0
        if mode == 2:
            p.bmin[rng.random(m) < 0.5] = -1e20
            p.bmax[rng.random(m) < 0.5] = 1e20
        if mode == 3:
            p.bmin *= 10
            p.bmax *= 10
        st = dict(eps_abs=float(rng.choice([1e-4, 1e-6, 1e-8])), eps_rel=float(rng.choice([1e-4, 1e-6, 1e-8])), verbose=0,
                  scaling=int(rng.choice([0, 1, 2, 10])), proximal=int(rng.integers(0, 2)), max_iter=int(rng.choice([50, 1000, 10000])),
                  sigma_init=float(rng.choice([2e1, 1.0, 1e3])), theta=float(rng.choice([0.25, 0.5])), delta=float(rng.choice([10, 100])),
                  gamma_init=float(rng.choice([1e1, 1e4, 1e7])), gamma_max=1e7, enable_dual_termination=int(rng.random() < 0.2),
                  factorization_method=int(rng.choice([0, 1, 1, 2])), inner_max_iter=int(rng.choice([5, 100])))
        warm = None
        if rng.random() < 0.3:
            warm = (rng.standard_normal(n), rng.standard_normal(m))
        if force:
            st.update(force)
        yield it, p, st, warm, dict(n=n, m=m, dA=dA, dM=dM, mode=mode)


def rel(a, b):
    return (np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0


def run_case(ctx, p, st, warm):
    """the engine and the oracle on one case -> dict(status, iter (engine, oracle), x / y relative differences)"""
    import oracle.binding as ob
    from qpalm_amd.solver import QpalmBatch
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    if warm is not None:
        bt.warm_start(warm[0][None, :], warm[1][None, :])
        o.warm_start(warm[0], warm[1])
    bt.solve()
    o.solve()
    info = bt.info(0)
    x, y = bt.solution()
    res = dict(status=(int(info.status_val), int(o.status_val)), iter=(int(info.iter), int(o.info.iter)),
               dx=rel(x[0], o.x), dy=rel(y[0], o.y), ymax=float(np.max(np.abs(o.y))) if o.y.size else 0.0)
    bt.close()
    o.cleanup()
    return res
