#!/bin/bash
# round 6, stage 4: the half-wave rank split of "table s-1 on the rows of block s" (QP_SPLIT_A) against the same build without it, same box
O=gpurun_out/r06_stage4; mkdir -p $O
tools/evidence/pl_probe > $O/permlane_probe.txt 2>&1; cat $O/permlane_probe.txt
timeout 900 python -m pytest tests/test_full_size.py tests/test_sweep32.py -q -m gpu -x > $O/pytest_sel.log 2>&1; tail -3 $O/pytest_sel.log
bash tools/evidence/gpu_ab.sh 8192 cur nosplit cur nosplit 2>&1 | tail -6
cp gpurun_out/gab.txt $O/ab_split_a.txt
