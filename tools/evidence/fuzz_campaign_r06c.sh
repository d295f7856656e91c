#!/bin/bash
# third fresh-seed campaign of round 6, on the build with the sparse factor's LDS forms: sparse factor only (natural ordering, nested dissection at
# n = 257..600 and at n = 600..1500, dual termination, nonconvex warm starts through the sparse factor), seeds no test and no earlier campaign uses
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r06/fuzz_final_c; mkdir -p $OUT
F=tools/evidence/fuzz_parity.py
timeout 600 python $F 941 400 hip 20 120 sparse=1 > $OUT/sparse_small_941.log 2>&1
timeout 600 python $F 942 80 hip 257 600 sparse=1 ordering=1 > $OUT/sparse_dissection_large_942.log 2>&1
timeout 600 python $F 943 300 hip 20 120 sparse=1 enable_dual_termination=1 > $OUT/sparse_dual_termination_943.log 2>&1
timeout 900 python $F 944 30 hip 600 1500 sparse=1 ordering=1 > $OUT/sparse_dissection_larger_944.log 2>&1
timeout 600 python $F 945 200 hip 20 120 sparse=1 ordering=1 enable_dual_termination=1 > $OUT/sparse_dissection_dual_945.log 2>&1
tail -q -n 1 $OUT/*.log
