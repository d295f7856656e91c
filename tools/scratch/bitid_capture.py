"""Bit-identity capture (development aid): solves a fixed set of QPs on the EMULATED kernels and writes x, y, the final factor
and the counters to an .npz; run before and after a change that must not alter any rounding, then compare with --compare."""
import sys, os, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qpalm_amd import build
from qpalm_amd.solver import Context, QpalmBatch
from qpalm_amd.problems import random_qp, random_mpc_qp

def run(kkt=False):
    ctx = Context(0, lib_path=build.build_emu())
    ctx.set_option("coop", 0)
    out = {}
    shapes = [(40, 70, 0.2, 0.1), (97, 150, 0.08, 0.05), (130, 260, 0.05, 0.03), (200, 380, 0.03, 0.02), (300, 420, 0.02, 0.01)]
    for si, (n, m, dA, dM) in enumerate(shapes):
        probs = [random_qp(n, m, seed=4000 + 10 * si + k, density_A=dA, density_M=dM) for k in range(2)]
        st = ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
        if kkt:
            st.factorization_method = 0
        bt = QpalmBatch(ctx, probs, st)
        bt.solve()
        x, y = bt.solution()
        for k in range(len(probs)):
            Lf, Df = bt.factor_rows(n + m, k) if kkt else bt.factor(k)
            out["x_%d_%d" % (si, k)] = x[k].copy(); out["y_%d_%d" % (si, k)] = y[k].copy()
            out["L_%d_%d" % (si, k)] = np.tril(np.array(Lf), -1).copy(); out["D_%d_%d" % (si, k)] = np.array(Df).copy()
            out["it_%d_%d" % (si, k)] = np.array([bt.info(k).iter, bt.info(k).status_val])
        bt.close()
    return out

if __name__ == "__main__":
    path = sys.argv[1]
    kkt = "--kkt" in sys.argv
    res = run(kkt)
    if "--compare" in sys.argv:
        ref = np.load(path)
        bad = 0
        for k in ref.files:
            if not np.array_equal(ref[k], res[k]):
                bad += 1
                print("DIFF", k, float(np.max(np.abs(ref[k] - res[k]))))
        print("compared", len(ref.files), "arrays:", "ALL BIT-IDENTICAL" if not bad else "%d differ" % bad)
        sys.exit(1 if bad else 0)
    np.savez(path, **res)
    print("saved", path, len(res))
