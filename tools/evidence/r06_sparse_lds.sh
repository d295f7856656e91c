#!/bin/bash
# The LDS forms of the sparse factorisation and solves (round 6) against the HBM forms ("sparse_lds" = 0), same box: the GPU tests of the sparse
# factor, the two sparse batch workloads with and without, a short default line (the dense kernels must not have moved).
OUT=gpurun_out/r06/sparse_lds
mkdir -p $OUT
timeout 900 python -m pytest tests/test_sparse_factor.py -q -m gpu -s --timeout 200 > $OUT/sparse_factor_at_size.txt 2>&1
for w in banded blocks; do
  for lds in 1 0; do
    timeout 400 python bench.py --workload sparse-$w-2000 --opt sparse_lds=$lds --no-cpu > $OUT/bench_sparse_${w}_2000_lds$lds.json 2>> $OUT/bench.err
  done
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu --no-mpc > $OUT/bench_default_short.json 2>> $OUT/bench.err
tail -n 30 $OUT/sparse_factor_at_size.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/sparse_lds/bench_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(j["value"], 1), j.get("solve_stats", {}).get("ms_per_qp_in_kernel"))
    except Exception as e:
        print(f, "unreadable", e)
PY
