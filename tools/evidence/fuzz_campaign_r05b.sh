#!/bin/bash
# second fresh-seed campaign of round 5, on the round's LAST build (128-thread instance, nested-dissection ordering, one-launch coop sweep,
# tiled line-search sort): seeds no test and no earlier campaign uses; logs under gpurun_out/r05/fuzz_final/
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r05/fuzz_final; mkdir -p $OUT
F=tools/evidence/fuzz_parity.py
timeout 600 python $F 611 400 hip 2 70 > $OUT/general_small_611.log 2>&1
timeout 600 python $F 612 120 hip 70 256 > $OUT/general_mid_612.log 2>&1
timeout 600 python $F 613 80 hip 257 600 > $OUT/general_large_613.log 2>&1
timeout 600 python $F 631 300 hip 2 70 small_workgroups=2 > $OUT/instance128_small_631.log 2>&1
timeout 600 python $F 632 100 hip 70 250 small_workgroups=2 > $OUT/instance128_mid_632.log 2>&1
timeout 600 python $F 641 300 hip 20 120 sparse=1 ordering=1 > $OUT/sparse_dissection_small_641.log 2>&1
timeout 600 python $F 642 60 hip 257 600 sparse=1 ordering=1 > $OUT/sparse_dissection_large_642.log 2>&1
timeout 600 python $F 651 200 hip 2 70 linesearch_hbm=32 > $OUT/tiled_sort_651.log 2>&1
timeout 600 python $F 661 200 hip 2 70 nonconvex=1 q_shift=1.0 > $OUT/nonconvex_661.log 2>&1
tail -q -n 1 $OUT/*.log
# coop mode forced on single QPs of 130..600 variables under the reference's refactorise-or-update rule: the one-launch update sweep
# (two to five row chunks, ragged last blocks, downdates, first nonzero anywhere) on random patterns
timeout 900 python $F 681 150 hip 130 600 coop=1 coop_rank_threshold=-1 factorization_method=1 > $OUT/coop_sweep_681.log 2>&1
timeout 900 python $F 682 60 hip 130 600 coop=1 coop_rank_threshold=-1 factorization_method=1 nonconvex=1 q_shift=1.0 > $OUT/coop_sweep_nonconvex_682.log 2>&1
tail -q -n 1 $OUT/coop_sweep*.log
