#!/bin/bash
# same-box A/B on the mpc-160 workload (working tree vs .ab_prev/)
mkdir -p gpurun_out
python bench.py --workload mpc-160 --steps 5 --warmup 1 --no-cpu > gpurun_out/abm_new.json 2> gpurun_out/abm.err
(cd .ab_prev && python bench.py --workload mpc-160 --steps 5 --warmup 1 --no-cpu) > gpurun_out/abm_prev.json 2>> gpurun_out/abm.err
tail -2 gpurun_out/abm.err
