#!/bin/bash
# after the sparse factor's default went to eight columns per wavefront (host side only): the GPU tests of the sparse factor and campaign S again,
# the sparse batch lines and the sparse factor at size into final/, then the PMC passes so that the summaries carry the tree's hashes
REPO=${GRAFT_REPO_ROOT:-$PWD}; cd $REPO
OUT=gpurun_out/r06/final; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sparse_factor.py tests/test_fuzz_seeds.py tests/test_qps.py -q -m gpu -k "sparse or qps" --timeout 600 > $OUT/pytest_gpu_sparse_after_gpw8.log 2>&1
timeout 600 python -m pytest tests/test_sparse_factor.py -q -m gpu -s --timeout 150 > $OUT/sparse_factor_at_size.txt 2>&1
timeout 400 python bench.py --workload sparse-banded-2000 > $OUT/bench_sparse_banded_2000.json 2>> $OUT/bench_default.err
timeout 400 python bench.py --workload sparse-blocks-2000 > $OUT/bench_sparse_blocks_2000.json 2>> $OUT/bench_default.err
bash tools/evidence/pmc_refresh.sh r06
tail -n 4 $OUT/pytest_gpu_sparse_after_gpw8.log
