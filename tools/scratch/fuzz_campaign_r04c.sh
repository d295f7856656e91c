#!/bin/bash
# third fresh-seed campaign of round 4 (run on the GPU box): the nonconvex front-end (LOBPCG + per-QP gamma) with indefinite Hessians
# (q_shift: the diagonal of Q lowered by its mean) and dual-objective termination forced; logs under gpurun_out/r04/fuzz/
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r04/fuzz; mkdir -p $OUT
timeout 900 python tools/evidence/fuzz_parity.py 271 400 hip 2 70 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_small_271.log 2>&1
timeout 900 python tools/evidence/fuzz_parity.py 272 150 hip 70 256 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_mid_272.log 2>&1
timeout 1200 python tools/evidence/fuzz_parity.py 273 60 hip 257 420 nonconvex=1 q_shift=0.5 > $OUT/nonconvex_large_273.log 2>&1
timeout 900 python tools/evidence/fuzz_parity.py 281 400 hip 2 70 enable_dual_termination=1 > $OUT/dual_termination_small_281.log 2>&1
timeout 900 python tools/evidence/fuzz_parity.py 282 100 hip 257 420 enable_dual_termination=1 > $OUT/dual_termination_large_282.log 2>&1
tail -q -n 1 $OUT/nonconvex_*.log $OUT/dual_termination_*.log
