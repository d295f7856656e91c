#!/bin/bash
# The PMC passes of round_artifacts.sh alone (headline: FETCH / WRITE / SQ; mpc-160 Schur and KKT: FETCH / WRITE), their summaries and the reported line:
# after a source change that does not touch what the kernels execute, so that the summaries carry the tree's hashes.  bash tools/evidence/pmc_refresh.sh [round]
R=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$R/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --no-cpu --no-mpc > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-mpc > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-mpc > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-mpc > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_mpc_fetch -- python3 $REPO/bench.py --workload mpc-160 --steps 5 --no-cpu > $OUT/pmc_mpc_fetch_bench.json 2> $OUT/pmc_mpc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_mpc_write -- python3 $REPO/bench.py --workload mpc-160 --steps 5 --no-cpu > $OUT/pmc_mpc_write_bench.json 2> $OUT/pmc_mpc_write.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_mpck_fetch -- python3 $REPO/bench.py --workload mpc-160 --kkt --batch 2048 --steps 3 --no-cpu > $OUT/pmc_mpck_fetch_bench.json 2> $OUT/pmc_mpck_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_mpck_write -- python3 $REPO/bench.py --workload mpc-160 --kkt --batch 2048 --steps 3 --no-cpu > $OUT/pmc_mpck_write_bench.json 2> $OUT/pmc_mpck_write.err
cd $OUT
find kt -name "*kernel_stats*.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_bench_default.csv \;
python3 $REPO/tools/evidence/pmc_summary.py $OUT $REPO > $OUT/k_solve_pmc_traffic.json
python3 $REPO/tools/evidence/pmc_summary.py $OUT $REPO mpc_ > $OUT/mpc160_pmc_traffic.json
python3 $REPO/tools/evidence/pmc_summary.py $OUT $REPO mpck_ > $OUT/mpc160_kkt_pmc_traffic.json
rm -rf kt pmc_fetch pmc_write pmc_sq pmc_mpc_fetch pmc_mpc_write pmc_mpck_fetch pmc_mpck_write
cd $REPO
timeout 1200 python bench.py --traffic-json $OUT/k_solve_pmc_traffic.json > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/k_solve_pmc_traffic.json
