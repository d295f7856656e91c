"""one fuzz case under the three pivot-sum modes: python tools/evidence/fuzz_case_modes.py <seed> <case> <n_lo> <n_hi> key=value ..."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch; torch.cuda.init()
from qpalm_amd.solver import Context
from tests.fuzz_cases import cases, run_case
pos = [a for a in sys.argv[1:] if "=" not in a]
force = {}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        force[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
seed, case, nlo, nhi = int(pos[0]), int(pos[1]), int(pos[2]), int(pos[3])
ctx = Context(0)
for mode in (-1, 1, 0):
    ctx.set_option("sequential_rank_sums", mode)
    for it, p, st, warm, meta in cases(seed, case + 1, nlo, nhi, force):
        if it != case:
            continue
        r = run_case(ctx, p, st, warm)
        print("mode", mode, meta, "status", r["status"], "iter", r["iter"], "dx", r["dx"], "dy", r["dy"], "guard", r["guard"], "obj", r["obj"], flush=True)
