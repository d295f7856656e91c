#!/bin/bash
# third fresh-seed campaign of round 5 (final build): the corners the first two did not reach -- nonconvex in KKT mode and on the sparse
# factor, dual termination beyond n = 70, and factors of more than 2048 rows (the large-factor sweep).  Logs under gpurun_out/r05/fuzz_final/
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r05/fuzz_final; mkdir -p $OUT
F=tools/evidence/fuzz_parity.py
timeout 600 python $F 691 150 hip 130 600 factorization_method=0 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_kkt_691.log 2>&1
timeout 600 python $F 692 150 hip 70 400 enable_dual_termination=1 > $OUT/dual_termination_mid_692.log 2>&1
timeout 600 python $F 693 100 hip 130 600 sparse=1 nonconvex=1 q_shift=1.0 > $OUT/sparse_nonconvex_693.log 2>&1
timeout 900 python $F 694 10 hip 2100 2500 factorization_method=1 > $OUT/large_factor_694.log 2>&1
timeout 900 python $F 695 6 hip 2100 2500 factorization_method=1 nonconvex=1 q_shift=1.0 > $OUT/large_factor_nonconvex_695.log 2>&1
tail -q -n 1 $OUT/nonconvex_kkt_691.log $OUT/dual_termination_mid_692.log $OUT/sparse_nonconvex_693.log $OUT/large_factor_694.log $OUT/large_factor_nonconvex_695.log
