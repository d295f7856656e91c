#!/bin/bash
# second fresh-seed campaign of round 6 on the final build: the size ranges and modes the first one left out
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r06/fuzz_final; mkdir -p $OUT
F=tools/evidence/fuzz_parity.py
timeout 900 python $F 911 60 hip 600 1100 > $OUT/general_xlarge_911.log 2>&1
timeout 600 python $F 912 150 hip 70 256 factorization_method=0 > $OUT/kkt_mid_912.log 2>&1
timeout 900 python $F 913 60 hip 257 420 nonconvex=1 q_shift=0.5 > $OUT/nonconvex_large_913.log 2>&1
timeout 600 python $F 914 200 hip 70 256 small_workgroups=0 > $OUT/general_mid_on_512_threads_914.log 2>&1
timeout 600 python $F 915 200 hip 20 120 sparse=1 ordering=1 enable_dual_termination=1 > $OUT/sparse_dissection_dual_915.log 2>&1
timeout 900 python $F 916 40 hip 1025 1400 factorization_method=1 > $OUT/general_k4_916.log 2>&1
timeout 600 python $F 917 200 hip 2 70 sequential_rank_sums=1 > $OUT/general_running_pivot_917.log 2>&1
tail -q -n 1 $OUT/general_xlarge_911.log $OUT/kkt_mid_912.log $OUT/nonconvex_large_913.log $OUT/general_mid_on_512_threads_914.log $OUT/sparse_dissection_dual_915.log $OUT/general_k4_916.log $OUT/general_running_pivot_917.log
