"""Randomised parity fuzzing: small random QPs with random shapes, bound patterns, settings (scaling, proximal, sigma, gamma,
dual termination, KKT / Schur, inner_max_iter, max_iter) and warm starts, the engine against the oracle: status and iteration
counts exact, x and y to 1e-8.  TEST TOOL (uses oracle/): python tools/fuzz_parity.py <seed> <cases> [hip|emu] [n_lo n_hi]."""
import sys, numpy as np, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import oracle.binding as ob
from qpalm_amd.problems import random_qp
from qpalm_amd.solver import Context, QpalmBatch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
backend = sys.argv[3] if len(sys.argv) > 3 else 'hip'
ctx = Context(0, lib_path=os.path.join(ROOT, 'tests', 'emu', 'libqpalm_gfx950_emu.so')) if backend == 'emu' else Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
NLO, NHI = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (2, 70)   # range of n (m up to 1.7 n)
rel = lambda a, b: (np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0
bad = 0    # status or iteration count differs (or an exception)
soft = 0   # same status and iterations, x or y beyond 1e-8 (seen only with sigma_init = 1e3, mostly in KKT mode: y = y + sigma (Ax - z)
           # amplifies the last bits of Ax by up to sigma_max = 1e9)
t0 = time.time()
for it in range(N):
    n = int(rng.integers(NLO, NHI)); m = int(rng.integers(1, max(2, int(1.7 * NHI))))
    dA = float(rng.choice([0.05, 0.15, 0.4, 1.0])) * min(1.0, 70.0 / n); dM = float(rng.choice([0.02, 0.1, 0.5])) * min(1.0, 70.0 / n)
    p = random_qp(n, m, seed=int(rng.integers(1 << 30)), density_A=dA, density_M=dM)
    # widen / tighten / equality / infinite bounds
    mode = rng.integers(0, 4)
    if mode == 1: p.bmax[:] = p.bmin + 0.0   # equalities
    if mode == 2: p.bmin[rng.random(m) < 0.5] = -1e20; p.bmax[rng.random(m) < 0.5] = 1e20
    if mode == 3: p.bmin *= 10; p.bmax *= 10
    st = dict(eps_abs=float(rng.choice([1e-4, 1e-6, 1e-8])), eps_rel=float(rng.choice([1e-4, 1e-6, 1e-8])), verbose=0,
              scaling=int(rng.choice([0, 1, 2, 10])), proximal=int(rng.integers(0, 2)), max_iter=int(rng.choice([50, 1000, 10000])),
              sigma_init=float(rng.choice([2e1, 1.0, 1e3])), theta=float(rng.choice([0.25, 0.5])), delta=float(rng.choice([10, 100])),
              gamma_init=float(rng.choice([1e1, 1e4, 1e7])), gamma_max=1e7, enable_dual_termination=int(rng.random() < 0.2),
              factorization_method=int(rng.choice([0, 1, 1, 2])), inner_max_iter=int(rng.choice([5, 100])))
    try:
        bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
        if rng.random() < 0.3:
            x0, y0 = rng.standard_normal(n), rng.standard_normal(m)
            bt.warm_start(x0[None, :], y0[None, :]); o.warm_start(x0, y0)
        bt.solve(); o.solve()
        info = bt.info(0)
        x, y = bt.solution()
        hard_ok = int(info.status_val) == o.status_val and int(info.iter) == int(o.info.iter)
        ok = hard_ok
        if ok and o.status_val in (1, 2):
            ok = rel(x[0], o.x) <= 1e-8 and rel(y[0], o.y) <= 1e-8
        if not ok:
            if hard_ok: soft += 1
            else: bad += 1
            print("MISMATCH" if not hard_ok else "SOFT", it, n, m, dA, dM, mode, st, "status", info.status_val, o.status_val, "iter", info.iter, o.info.iter,
                  "x", rel(x[0], o.x), "y", rel(y[0], o.y))
    except Exception as e:
        bad += 1
        print("EXC", it, n, m, st, repr(e)[:300])
print("done", N, "status/iteration mismatches", bad, "x/y beyond 1e-8", soft, "time", round(time.time() - t0, 1))
