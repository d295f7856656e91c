#!/usr/bin/env python3
"""one fuzz case under the engine's modes: coop_case.py seed case n_lo n_hi [key=value settings ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from qpalm_amd.solver import Context  # noqa: E402
from tests.fuzz_cases import cases, run_case  # noqa: E402

seed, want, nlo, nhi = (int(a) for a in sys.argv[1:5])
force = {}
for a in sys.argv[5:]:
    k, v = a.split("=", 1)
    force[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
ctx = Context(0, lib_path=os.environ.get("QPALM_LIB") or None)
only_one = bool(os.environ.get("QPALM_LIB"))
for it, p, st, warm, meta in cases(seed, want + 1, nlo, nhi, force or None):
    if it != want:
        continue
    print(meta, {k: st[k] for k in ("factorization_method", "sigma_init", "scaling", "proximal", "max_iter")})
    for name, opts in (("one workgroup", dict(coop=0)), ("coop, chain of launches", dict(coop=1, coop_updates=1, coop_rank_threshold=-1)),
                       ("coop, one-launch sweep", dict(coop=1, coop_updates=2, coop_rank_threshold=-1)), ("coop, cost rule", dict(coop=1, coop_updates=2, coop_rank_threshold=-2))):
        if only_one and name != "one workgroup":
            continue
        for k, v in opts.items():
            ctx.set_option(k, v)
        r = run_case(ctx, p, st, warm)
        print("%-26s status %s iter %s dx %.3e dy %.3e obj %s" % (name, r["status"], r["iter"], r["dx"], r["dy"], r["obj"]))
        sys.stdout.flush()
