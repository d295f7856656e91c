#!/bin/bash
# Round evidence, run on the GPU box:  bash tools/evidence/round_artifacts.sh [round]   (outputs under gpurun_out/<round>/final/)
# Copies to profiles/<round>/ are made by hand from the merged gpurun_out/.  Every step has its own timeout.
R=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$R/final
mkdir -p $OUT
cd $REPO
timeout 1800 python -m pytest tests -q -m gpu --timeout 400 > $OUT/pytest_gpu.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp
# kernel trace (own run) and the PMC passes (separate runs, counters only), all of the default bench command
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --no-cpu --no-mpc > $OUT/bench_under_kernel_trace.json 2> $OUT/kt.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-mpc > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-mpc > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu --no-mpc > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktm -- python3 $REPO/bench.py --workload mpc-160 --steps 5 --no-cpu > $OUT/bench_mpc160_under_kernel_trace.json 2> $OUT/ktm.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktk -- python3 $REPO/bench.py --workload mpc-160 --kkt --batch 2048 --steps 3 --no-cpu > $OUT/bench_mpc160_kkt_under_kernel_trace.json 2> $OUT/ktk.err
# config 3 (mpc-160), HBM bytes per warm-started launch: the same two counters in passes of their own (round 6: the roofline object of the mpc160 lines)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_mpc_fetch -- python3 $REPO/bench.py --workload mpc-160 --steps 5 --no-cpu > $OUT/pmc_mpc_fetch_bench.json 2> $OUT/pmc_mpc_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_mpc_write -- python3 $REPO/bench.py --workload mpc-160 --steps 5 --no-cpu > $OUT/pmc_mpc_write_bench.json 2> $OUT/pmc_mpc_write.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_mpck_fetch -- python3 $REPO/bench.py --workload mpc-160 --kkt --batch 2048 --steps 3 --no-cpu > $OUT/pmc_mpck_fetch_bench.json 2> $OUT/pmc_mpck_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_mpck_write -- python3 $REPO/bench.py --workload mpc-160 --kkt --batch 2048 --steps 3 --no-cpu > $OUT/pmc_mpck_write_bench.json 2> $OUT/pmc_mpck_write.err
cd $OUT
find kt -name "*kernel_stats*.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_bench_default.csv \;
find ktm -name "*kernel_stats*.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_bench_mpc160.csv \;
find ktk -name "*kernel_stats*.csv" -exec cp {} $OUT/rocprofv3_kernel_stats_bench_mpc160_kkt.csv \;
python3 $REPO/tools/evidence/pmc_summary.py $OUT $REPO > $OUT/k_solve_pmc_traffic.json
python3 $REPO/tools/evidence/pmc_summary.py $OUT $REPO mpc_ > $OUT/mpc160_pmc_traffic.json
python3 $REPO/tools/evidence/pmc_summary.py $OUT $REPO mpck_ > $OUT/mpc160_kkt_pmc_traffic.json
rm -rf kt ktm ktk pmc_fetch pmc_write pmc_sq pmc_mpc_fetch pmc_mpc_write pmc_mpck_fetch pmc_mpck_write
cd $REPO
# the reported line: same build, same box, traffic from the passes above (bench.py checks the source and library hashes recorded in the summary)
timeout 1200 python bench.py --traffic-json $OUT/k_solve_pmc_traffic.json > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --batch 512 --no-cpu --no-mpc > $OUT/bench_b512.json 2>> $OUT/bench_default.err
timeout 600 python bench.py --workload mpc-160 --steps 5 > $OUT/bench_mpc160.json 2>> $OUT/bench_default.err
timeout 600 python bench.py --workload mpc-160 --kkt --batch 2048 --steps 3 --no-cpu > $OUT/bench_mpc160_kkt.json 2>> $OUT/bench_default.err
timeout 600 python bench.py --workload mpc-160 --steps 5 --small-workgroups 3 --no-cpu > $OUT/bench_mpc160_256_thread_instance.json 2>> $OUT/bench_default.err
timeout 600 python bench.py --sweep-ranks 32 --no-cpu --no-mpc > $OUT/bench_sweep_ranks_32.json 2>> $OUT/bench_default.err
# the price of the pivot guard, same box: default (guarded prefix tree) / unguarded tree / the running pivot in every column
for v in -1 0 1; do timeout 600 python bench.py --no-cpu --no-mpc --sequential-rank-sums $v > $OUT/bench_rank_sums_$v.json 2>> $OUT/bench_default.err; done
timeout 120 tools/evidence/mb_f64rate > $OUT/mb_f64rate.txt 2>&1
bash tools/evidence/phase_traffic.sh $R/final/phase_traffic > $OUT/phase_traffic.log 2>&1
timeout 300 python tools/evidence/sweep_probe.py cur --reps 4 --ranks 16 8 > $OUT/sweep_probe.txt 2>> $OUT/bench_default.err
timeout 300 python tools/evidence/setup_timing.py 8192 > $OUT/setup_timing.txt 2>&1
timeout 600 python tools/evidence/coop_timing.py 512 1000 2500 5000 > $OUT/coop_timing.txt 2>> $OUT/bench_default.err
# config 5 as written (n = 5000 nonconvex): wall time and the coop profile of one solve, every round
QPALM_COOP_PROFILE=1 timeout 600 python -m pytest tests/test_coop.py -q -m gpu -k config5 -s --timeout 300 > $OUT/config5.txt 2>&1
# batch sizes between "one QP" and "the chip is full" (missing #2)
mkdir -p $REPO/gpurun_out/$R/batch
for b in 16 64 128 256; do timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu --no-mpc > $REPO/gpurun_out/$R/batch/bench_b$b.json 2>> $OUT/bench_default.err; done
# the sparse factor at size
timeout 600 python -m pytest tests/test_sparse_factor.py -q -m gpu -s --timeout 150 > $OUT/sparse_factor_at_size.txt 2>&1
# batches of sparse QPs on the sparse factor (SURVEY section 8 row h), with the oracle's sparse-storage mode on the host cores beside them
timeout 400 python bench.py --workload sparse-banded-2000 > $OUT/bench_sparse_banded_2000.json 2>> $OUT/bench_default.err
timeout 400 python bench.py --workload sparse-blocks-2000 > $OUT/bench_sparse_blocks_2000.json 2>> $OUT/bench_default.err
timeout 300 tools/evidence/sload_coherence_test > $OUT/sload_coherence.txt 2>&1
timeout 200 tools/evidence/host_alloc_probe > $OUT/host_alloc_probe.txt 2>&1
ls -la $OUT
