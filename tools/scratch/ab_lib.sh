#!/bin/bash
# same-box A/B of two HIP builds on the headline workload: tools/scratch/ab_lib.sh <other lib> [bench args]
mkdir -p gpurun_out
OTHER=$1; shift
for rep in 1 2; do
for which in default other; do
  if [ $which = other ]; then L="--lib $OTHER"; else L=""; fi
  python bench.py --steps 1 --warmup 1 --no-cpu $L $@ > gpurun_out/abl_$which.json 2>> gpurun_out/abl.err
  python - <<PY
import json; d=json.load(open("gpurun_out/abl_$which.json")); p=d["solve_stats"]["phase_ms_per_qp"]; print("$which", round(d["value"]), round(d["roofline"]["frac"],3), "total", round(p["total"],2), "update", round(p["update"],2), "panel", round(p["dbg"][1],2), "trail", round(p["dbg"][2],2), "factor", round(p["factor"],2), "solve", round(p["solve"],2), d["solve_stats"]["all_solved"])
PY
done
done
tail -2 gpurun_out/abl.err
