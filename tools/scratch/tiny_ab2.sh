#!/bin/bash
# random QPs of three small sizes on the 256-thread instance vs the 128-thread one: same box
mkdir -p gpurun_out
for nm in "64 128" "128 256" "250 400"; do
  set -- $nm
  for sw in 1 2 1 2; do
    timeout 600 python bench.py --n $1 --m $2 --batch 16384 --small-workgroups $sw --steps 3 --warmup 1 --no-cpu --no-mpc > gpurun_out/tiny2_$sw.json 2>> gpurun_out/tiny2.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/tiny2_$sw.json").read().strip().splitlines()[-1])
print("n $1 m $2 small_workgroups $sw", round(d["value"]), "QP/s", round(d["ms_per_step"], 2), "ms/step", d["solve_stats"].get("all_solved"), d["solve_stats"].get("solution_sha256_16"))
PY
  done
done
tail -3 gpurun_out/tiny2.err
