"""TEST TOOL (CPU only): the ORACLE's own builds (plain, FMA, -Ofast, engine-form recurrence) over a whole fuzz family -- how often does a build of the reference's
restatement end MAX_ITER where another build of the same source solves?  The yardstick for the engine's stalls on ill-conditioned families (round 6: campaign 721,
Hessians scaled by 1e-10).  python tools/evidence/oracle_family_stalls.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from multiprocessing import Pool
def work(rng):
    lo, hi = rng
    from tests.fuzz_cases import cases, oracle_outcomes
    import oracle.binding as ob
    out = []
    for it, p, st, warm, meta in cases(721, 200, 70, 300, dict(q_scale=1e-10, factorization_method=1)):
        if it < lo or it >= hi: continue
        o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
        if warm is not None: o.warm_start(warm[0], warm[1])
        o.solve()
        plain = (int(o.status_val), int(o.info.iter)); o.cleanup()
        var = oracle_outcomes(p, st, warm)
        out.append((it, st["max_iter"], plain, var))
    return out
if __name__ == "__main__":
    t0 = time.time()
    with Pool(7) as pool:
        res = pool.map(work, [(k, k + 10) for k in range(0, 200, 10)])
    rows = [r for chunk in res for r in chunk]
    stall = {}
    for it, mi, plain, var in rows:
        allv = dict(var, plain=plain)
        sts = {k: v for k, v in allv.items()}
        if len({v[0] for v in allv.values()}) > 1:
            print("case", it, "max_iter", mi, allv)
        for k, v in allv.items():
            if v[0] == -2 and any(w[0] == 1 for w in allv.values()):
                stall[k] = stall.get(k, 0) + 1
    print("MAX_ITER where another build solves, per build:", stall, "cases", len(rows), "time", round(time.time() - t0))
