#!/bin/bash
# fresh-seed campaign of round 6 on the round's LAST build (per-column pivot guard, Newton-direction guard, sparse factor with changed penalties
# as path updates and with dual termination): seeds no test and no earlier campaign uses; logs under gpurun_out/r06/fuzz_final/
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r06/fuzz_final; mkdir -p $OUT
F=tools/evidence/fuzz_parity.py
timeout 600 python $F 811 400 hip 2 70 > $OUT/general_small_811.log 2>&1
timeout 600 python $F 812 120 hip 70 256 > $OUT/general_mid_812.log 2>&1
timeout 600 python $F 813 80 hip 257 600 > $OUT/general_large_813.log 2>&1
timeout 600 python $F 821 300 hip 2 70 factorization_method=0 sigma_init=1e3 > $OUT/kkt_sigma_821.log 2>&1
timeout 600 python $F 831 300 hip 2 70 small_workgroups=2 > $OUT/instance128_small_831.log 2>&1
timeout 600 python $F 841 300 hip 20 120 sparse=1 > $OUT/sparse_small_841.log 2>&1
timeout 600 python $F 842 60 hip 257 600 sparse=1 ordering=1 > $OUT/sparse_dissection_large_842.log 2>&1
timeout 600 python $F 843 200 hip 20 120 sparse=1 enable_dual_termination=1 > $OUT/sparse_dual_termination_843.log 2>&1
timeout 600 python $F 861 300 hip 2 70 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_small_861.log 2>&1
timeout 600 python $F 862 100 hip 70 256 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_mid_862.log 2>&1
timeout 600 python $F 871 300 hip 2 70 enable_dual_termination=1 > $OUT/dual_termination_871.log 2>&1
# linear programmes (Q = 0: pivots down to 1 / gamma_max, H exactly singular without the proximal term): the guards' home ground
timeout 900 python $F 711 200 hip 70 400 lp=1 factorization_method=1 > $OUT/lp_schur_711.log 2>&1
timeout 600 python $F 712 300 hip 2 70 lp=1 > $OUT/lp_small_712.log 2>&1
# positive-diagonal but tiny / rank-deficient Hessians (what round 5's structural hint missed): Q scaled by 1e-10
timeout 600 python $F 721 200 hip 70 300 q_scale=1e-10 factorization_method=1 > $OUT/tiny_q_721.log 2>&1
# coop mode forced (one-launch update sweep) under the reference's refactorise-or-update rule
timeout 900 python $F 881 100 hip 130 600 coop=1 coop_rank_threshold=-1 factorization_method=1 > $OUT/coop_sweep_881.log 2>&1
tail -q -n 1 $OUT/*.log
