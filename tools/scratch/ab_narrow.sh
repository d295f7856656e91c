#!/bin/bash
mkdir -p gpurun_out
for nr in 1 0 1 0; do
  python bench.py --workload mpc-160 --steps 5 --warmup 1 --no-cpu --narrow-rows $nr > gpurun_out/abn_$nr.json 2>> gpurun_out/abn.err
  python - <<PY
import json; d=json.load(open("gpurun_out/abn_$nr.json")); p=d["solve_stats"]["phase_ms_per_qp"]; print("narrow=$nr", round(d["value"]), "total", round(p["total"],3), "factor", round(p["factor"],3), "form", round(p["dbg"][3],3), "update", round(p["update"],3), "solve", round(p["solve"],3), "ls", round(p["linesearch"],3), "resid", round(p["residuals"],3), d["solve_stats"]["all_solved"])
PY
done
tail -2 gpurun_out/abn.err
