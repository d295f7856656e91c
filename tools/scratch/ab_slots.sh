#!/bin/bash
# panel-wave time per QP against the number of resident workgroups (interference check)
mkdir -p gpurun_out
for sl in 512 256 128; do
  python bench.py --steps 1 --warmup 0 --no-cpu --batch 1024 --max-slots $sl > gpurun_out/absl_$sl.json 2>> gpurun_out/absl.err
  python - <<PY
import json; d=json.load(open("gpurun_out/absl_$sl.json")); p=d["solve_stats"]["phase_ms_per_qp"]; print("slots=$sl", round(d["value"]), "total", round(p["total"],2), "update", round(p["update"],2), "panel", round(p["dbg"][1],2), "trail", round(p["dbg"][2],2), "factor", round(p["factor"],2), "solve", round(p["solve"],2), "ls", round(p["linesearch"],2), "resid", round(p["residuals"],2))
PY
done
tail -2 gpurun_out/absl.err
