"""Where the persistent workgroups run: histogram of QPGStats.placement over one batch (MI355X only)."""
import collections, sys
import numpy as np
from qpalm_amd.problems import random_qp
from qpalm_amd.solver import Context, QpalmBatch

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    place = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ctx = Context(0)
    ctx.set_option("place_panel_wave", place)
    base = [random_qp(n, 2 * n, seed=1000 + k) for k in range(8)]
    probs = [base[k % 8] for k in range(B)]
    s = ctx.default_settings(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    bt = QpalmBatch(ctx, probs, s)
    bt.solve()
    pl = np.array([st.placement for st in bt.stats_all()], dtype=np.int64)
    key, arr, pw, simd0 = pl >> 16, (pl >> 8) & 255, (pl >> 4) & 15, pl & 15
    wgs = {}
    for k, a, p, s0 in zip(key, arr, pw, simd0):
        wgs[(int(k), int(a))] = (int(p), int(s0))
    per_cu = collections.Counter(k for k, _ in wgs)
    print("launch shape", bt.launch_shape(), "distinct workgroups seen", len(wgs), "CUs", len(per_cu), "workgroups per CU", collections.Counter(per_cu.values()))
    print("SIMD of wavefront 0 by arrival index", collections.Counter((a, s0) for (_, a), (_, s0) in wgs.items()))
    print("panel wavefront by arrival index", collections.Counter((a, p) for (_, a), (p, _) in wgs.items()))
    print("XCC ids", sorted(set(int(k) >> 8 for k in key)), "se/sh/cu ids", len(set(int(k) & 255 for k in key)))

if __name__ == "__main__":
    main()
