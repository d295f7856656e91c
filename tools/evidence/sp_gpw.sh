#!/bin/bash
# columns per wavefront of the sparse factorisation (context option sparse_gpw): 4 (groups of 16 lanes) against 8 (groups of 8) and 2, same box
OUT=gpurun_out/r06/sp_gpw; mkdir -p $OUT
for g in 4 8 2; do
  for w in banded blocks; do
    timeout 300 python bench.py --workload sparse-$w-2000 --opt sparse_gpw=$g --no-cpu > $OUT/gpw${g}_$w.json 2>> $OUT/err.txt
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/sp_gpw/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1]); k = j["solve_stats"]["ms_per_qp_in_kernel"]
        print("%-22s %8.1f QP/s  factor %.2f solve %.2f update %.2f total %.2f" % (f.split("/")[-1], j["value"], k["factor"], k["solve"], k["update"], k["total"]))
    except Exception as e:
        print(f, "unreadable", e)
PY
