#!/bin/bash
# same-box A/B of two builds: the working tree against a copy of an older commit under .ab_prev/ (git archive <rev> | tar x -C .ab_prev,
# built there; it ships to the GPU box with the snapshot).  tools/scratch/ab.sh [batch] -> gpurun_out/ab_{new,prev}.json
B=${1:-4096}
mkdir -p gpurun_out
python bench.py --batch $B --steps 2 --warmup 1 --no-cpu > gpurun_out/ab_new.json 2> gpurun_out/ab.err
(cd .ab_prev && python bench.py --batch $B --steps 2 --warmup 1 --no-cpu) > gpurun_out/ab_prev.json 2>> gpurun_out/ab.err
python bench.py --batch $B --steps 2 --warmup 1 --no-cpu > gpurun_out/ab_new2.json 2>> gpurun_out/ab.err
tail -2 gpurun_out/ab.err
