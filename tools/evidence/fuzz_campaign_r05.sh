#!/bin/bash
# fresh-seed campaigns of round 5 (run on the GPU box): seeds no test uses; logs under gpurun_out/r05/fuzz/
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r05/fuzz; mkdir -p $OUT
F=tools/evidence/fuzz_parity.py
timeout 900 python $F 501 500 hip 2 70 > $OUT/general_small_501.log 2>&1
timeout 900 python $F 502 150 hip 70 256 > $OUT/general_mid_502.log 2>&1
timeout 900 python $F 503 100 hip 257 600 > $OUT/general_large_503.log 2>&1
timeout 900 python $F 521 300 hip 2 70 factorization_method=0 sigma_init=1e3 > $OUT/kkt_sigma1e3_521.log 2>&1
timeout 900 python $F 571 300 hip 2 70 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_small_571.log 2>&1
timeout 900 python $F 572 100 hip 70 256 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_mid_572.log 2>&1
timeout 900 python $F 581 300 hip 2 70 enable_dual_termination=1 > $OUT/dual_termination_581.log 2>&1
timeout 900 python $F 591 400 hip 2 70 sparse=1 > $OUT/sparse_small_591.log 2>&1
timeout 900 python $F 592 100 hip 257 600 sparse=1 > $OUT/sparse_large_592.log 2>&1
tail -q -n 1 $OUT/*.log
