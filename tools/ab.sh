#!/bin/bash
# same-box A/B of two HIP builds: tools/ab.sh [batch] -> gpurun_out/ab_{new,prev}.json   (prev = qpalm_amd/lib/libqpalm_gfx950_prev.so)
B=${1:-512}
python bench.py --batch $B --steps 2 --warmup 1 --no-cpu > gpurun_out/ab_new.json 2> gpurun_out/ab.err
python bench.py --lib $PWD/qpalm_amd/lib/libqpalm_gfx950_prev.so --batch $B --steps 2 --warmup 1 --no-cpu > gpurun_out/ab_prev.json 2>> gpurun_out/ab.err
tail -2 gpurun_out/ab.err
