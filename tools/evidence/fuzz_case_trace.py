"""where a solve stalls: python tools/evidence/fuzz_case_trace.py <seed> <case> <n_lo> <n_hi> <mode> <small_workgroups> key=value ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch; torch.cuda.init()
import numpy as np
from qpalm_amd.solver import Context, QpalmBatch
from tests.fuzz_cases import cases
pos = [a for a in sys.argv[1:] if "=" not in a]
force = {}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        force[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
seed, case, nlo, nhi, mode, sw = [int(x) for x in pos[:6]]
ctx = Context(0)
ctx.set_option("sequential_rank_sums", mode); ctx.set_option("small_workgroups", sw)
import oracle.binding as ob
for it, p, st, warm, meta in cases(seed, case + 1, nlo, nhi, force):
    if it != case:
        continue
    print(meta, st)
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    if warm is not None: o.warm_start(warm[0], warm[1])
    o.enable_trace(3000); o.solve(); t = o.trace()
    print("oracle", o.status_val, int(o.info.iter), "outer", int(o.info.iter_out), "obj", float(o.info.objective))
    bt = QpalmBatch(ctx, [p], ctx.default_settings(**st))
    if warm is not None: bt.warm_start(warm[0][None, :], warm[1][None, :])
    bt.begin_solve()
    k = 0
    while k < 1500:
        step = 1 if k < 40 else 25
        bt.iterate(step); k += step
        i, s = bt.info(0), bt.stats(0)
        ko = min(k, len(t["kind"])) - 1
        print("it %5d kind %d outer %3d pri %.3e dua %.3e dua2 %.3e tau %.3e gamma %.1e act %4d ent %3d lea %3d rank1 %6d refac %4d seqcols %5d | oracle pri %.3e dua2 %.3e act %4d" % (
            int(i.iter), int(s.last_kind), int(i.iter_out), float(i.pri_res_norm), float(i.dua_res_norm), float(i.dua2_res_norm), float(s.tau), float(s.gamma), int(s.nb_active), int(s.nb_enter), int(s.nb_leave),
            int(s.n_rank1), int(s.n_refactor), int(s.n_seq_columns), float(t["pri_res_norm"][ko]) if "pri_res_norm" in t else -1, float(t["dua2_res_norm"][ko]), int(t["nb_active"][ko])), flush=True)
        if int(i.status_val) != -10:
            print("status", int(i.status_val)); break
