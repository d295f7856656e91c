#!/bin/bash
# same-box A/B of the 256-thread instance on the mpc-160 workload (and a check that random-1000 is unchanged)
mkdir -p gpurun_out
for sw in 1 0 1 0; do
  python bench.py --workload mpc-160 --steps 5 --warmup 1 --no-cpu --small-workgroups $sw > gpurun_out/abs_$sw.json 2>> gpurun_out/abs.err
  python - <<PY
import json; d=json.load(open("gpurun_out/abs_$sw.json")); print("small=$sw", round(d["value"]), d["config"].get("workgroups"), d["roofline"]["phases"]["solve"]["ms_per_qp"], d["roofline"]["phases"]["update"]["ms_per_qp"], d["roofline"]["phases"]["factor"]["ms_per_qp"])
PY
done
tail -2 gpurun_out/abs.err
