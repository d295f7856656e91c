#!/bin/bash
# same-box A/B of the panel-wave SIMD placement on the headline workload
mkdir -p gpurun_out
for pw in 1 0 1 0; do
  python bench.py --steps 1 --warmup 1 --no-cpu --place-panel-wave $pw > gpurun_out/abp_$pw.json 2>> gpurun_out/abp.err
  python - <<PY
import json; d=json.load(open("gpurun_out/abp_$pw.json")); p=d["solve_stats"]["phase_ms_per_qp"]; print("place=$pw", round(d["value"]), round(d["roofline"]["frac"],3), "update", round(p["update"],2), "panel", round(p["dbg"][1],2), "factor", round(p["factor"],2), "solve", round(p["solve"],2))
PY
done
tail -2 gpurun_out/abp.err
