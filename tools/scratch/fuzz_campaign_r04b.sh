#!/bin/bash
# second fresh-seed campaign of round 4 (run on the GPU box): the size ranges the first one and the suite leave out --
# n = 70..256 (the 256-thread instance with several block columns, KKT panels of up to ~700 rows on the 512-thread instance)
# and n = 600..1100 (two and four rows per thread in the update sweep); logs under gpurun_out/r04/fuzz/
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r04/fuzz; mkdir -p $OUT
for s in 241 242; do timeout 900 python tools/evidence/fuzz_parity.py $s 300 hip 70 256 > $OUT/general_mid_$s.log 2>&1; done
timeout 900 python tools/evidence/fuzz_parity.py 243 200 hip 70 256 factorization_method=0 > $OUT/kkt_mid_243.log 2>&1
timeout 900 python tools/evidence/fuzz_parity.py 244 200 hip 70 256 factorization_method=0 sigma_init=1e3 > $OUT/kkt_sigma1e3_mid_244.log 2>&1
timeout 1500 python tools/evidence/fuzz_parity.py 251 80 hip 600 1100 factorization_method=1 > $OUT/schur_large_251.log 2>&1
timeout 1500 python tools/evidence/fuzz_parity.py 252 40 hip 600 1100 > $OUT/general_xlarge_252.log 2>&1
tail -q -n 1 $OUT/general_mid_*.log $OUT/kkt_mid_*.log $OUT/kkt_sigma1e3_mid_*.log $OUT/schur_large_*.log $OUT/general_xlarge_*.log
