#!/bin/bash
# fresh-seed fuzz campaign of round 4 (run on the GPU box): seeds no test uses; logs under gpurun_out/r04/fuzz/
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r04/fuzz; mkdir -p $OUT
for s in 201 202 203 204; do timeout 600 python tools/evidence/fuzz_parity.py $s 500 hip 2 70 > $OUT/general_small_$s.log 2>&1; done
for s in 211 212; do timeout 900 python tools/evidence/fuzz_parity.py $s 200 hip 257 600 > $OUT/general_large_$s.log 2>&1; done
for s in 221 222; do timeout 600 python tools/evidence/fuzz_parity.py $s 400 hip 2 70 factorization_method=0 sigma_init=1e3 > $OUT/kkt_sigma1e3_$s.log 2>&1; done
timeout 900 python tools/evidence/fuzz_parity.py 231 100 hip 257 420 factorization_method=0 > $OUT/kkt_large_231.log 2>&1
tail -q -n 1 $OUT/*.log
