#!/usr/bin/env python3
"""Is a fuzz case's iteration count a property of the algorithm?  Runs the ORACLE on one case of tests/fuzz_cases.py four ways:
as built (-O2, no contraction), -O0, -O3 -ffp-contract=fast -mfma, -Ofast -mfma.  A case on which these disagree is one where
rounding decides the count (or the status); a GPU-vs-oracle mismatch there says nothing about either implementation.
TEST TOOL (uses oracle/).  usage: oracle_stability.py seed case n_lo n_hi [key=value ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.binding as ob  # noqa: E402
from tests.fuzz_cases import cases  # noqa: E402

pos = [a for a in sys.argv[1:] if "=" not in a]
force = {}
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=", 1)
        force[k] = float(v) if ("." in v or "e" in v.lower()) else int(v)
seed, want, lo, hi = int(pos[0]), int(pos[1]), int(pos[2]), int(pos[3])
libs = {}
for name, flags in (("O0", ["-O0", "-ffp-contract=off"]), ("fma", ["-O3", "-ffp-contract=fast", "-mfma"]), ("Ofast", ["-Ofast", "-mfma"])):
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "libqpalm_oracle_%s_%d.so" % (name, os.getpid()))
    subprocess.check_call(["gcc", "-std=c99", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "oracle", "qpalm_oracle.c"), "-lm"] + flags)
    libs[name] = out
try:
    for it, p, st, warm, meta in cases(seed, want + 1, lo, hi, force or None):
        if it != want:
            continue
        res = {}
        for name, lib in [("as built", None)] + list(libs.items()):
            kw = dict(settings=ob.default_settings(**st))
            if lib:
                kw["libpath"] = lib
            o = ob.OracleQP(*p.args(), **kw)
            if warm is not None:
                o.warm_start(warm[0], warm[1])
            o.solve()
            res[name] = (o.status_val, int(o.info.iter), int(o.info.iter_out))
            o.cleanup()
        print("seed %d case %d n=%d m=%d fm=%s sigma_init=%g: (status, iter, iter_out) %s" % (
            seed, want, meta["n"], meta["m"], st.get("factorization_method"), st.get("sigma_init", 20.0), res))
finally:
    for f in libs.values():
        os.remove(f)
