#!/bin/bash
# same-box A/B of builds of the sparse factor (libraries under .ab_libs/): the two sparse batch workloads per build
OUT=gpurun_out/r06/sp_variants
mkdir -p $OUT
for lib in .ab_libs/lib_*.so; do
  n=$(basename $lib .so)
  for w in banded blocks; do
    timeout 300 python bench.py --workload sparse-$w-2000 --lib $lib --no-cpu > $OUT/${n}_$w.json 2>> $OUT/err.txt
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/sp_variants/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        k = j["solve_stats"]["ms_per_qp_in_kernel"]
        print("%-28s %8.1f QP/s  factor %.2f solve %.2f update %.2f total %.2f" % (f.split("/")[-1], j["value"], k["factor"], k["solve"], k["update"], k["total"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
