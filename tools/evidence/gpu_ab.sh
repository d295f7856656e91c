#!/bin/bash
# same-box comparison of HIP builds with the phase split: tools/evidence/gpu_ab.sh <batch> <name> [<name> ...]
# (name = suffix of qpalm_amd/lib/libqpalm_gfx950_<name>.so; "cur" = the shipped build)   -> gpurun_out/gab_<name>.json, gpurun_out/gab.txt
mkdir -p gpurun_out
B=$1; shift
for name in "$@"; do
  lib=$PWD/qpalm_amd/lib/libqpalm_gfx950_$name.so
  [ "$name" = "cur" ] && lib=$PWD/qpalm_amd/lib/libqpalm_gfx950.so
  timeout 900 python bench.py --lib $lib --batch $B --steps 2 --warmup 1 --no-cpu > gpurun_out/gab_$name.json 2>> gpurun_out/gab.err
  python - <<PY | tee -a gpurun_out/gab.txt
import json
try:
    d = json.loads(open("gpurun_out/gab_$name.json").read().strip().splitlines()[-1]); p = d["solve_stats"]["phase_ms_per_qp"]; g = p["dbg"]
    print("$name B=$B", round(d["value"]), "frac", round(d["roofline"]["frac"], 3), "total", round(p["total"], 2), "update", round(p["update"], 2), "panel", round(g[1], 2),
          "trail", round(g[2], 2), "sweepwall", round(g[7], 2), "a/b/c/bar", [round(g[k], 2) for k in (8, 9, 10, 11)], "factor", round(p["factor"], 2), "solve", round(p["solve"], 2),
          "ls", round(p["linesearch"], 2), "res", round(p["residuals"], 2), d["solve_stats"]["all_solved"], d["solve_stats"].get("solution_sha256_16"), "copy", round(d["roofline"]["measured_copy_GBps"]))
except Exception as e:
    print("$name FAILED", e)
PY
done
tail -3 gpurun_out/gab.err
