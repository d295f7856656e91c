#!/bin/bash
# mpc-160 on the 256-thread instance (four workgroups per CU) vs the 128-thread one (seven): same box
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_parity.py -m gpu -q -k "workgroup_shapes" 2>&1 | tail -3
for sw in 1 2 1 2; do
  for B in 8192 16384; do
    timeout 600 python bench.py --workload mpc-160 --batch $B --small-workgroups $sw --steps 5 --warmup 2 --no-cpu > gpurun_out/tiny_$sw.json 2>> gpurun_out/tiny.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/tiny_$sw.json").read().strip().splitlines()[-1])
print("small_workgroups $sw B=$B", round(d["value"]), "QP/s", d["ms_per_step"], "ms/step", d["solve_stats"].get("all_solved"), d["solve_stats"].get("solution_sha256_16"))
PY
  done
done
tail -3 gpurun_out/tiny.err
