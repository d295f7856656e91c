#!/bin/bash
# same-box comparison of several HIP builds: tools/scratch/ab_multi.sh <lib> [<lib> ...]  (the default library first and last)
mkdir -p gpurun_out
run() { python bench.py --batch 4096 --steps 1 --warmup 1 --no-cpu $2 > gpurun_out/abx.json 2>> gpurun_out/abx.err; python - <<PY
import json; d=json.load(open("gpurun_out/abx.json")); p=d["solve_stats"]["phase_ms_per_qp"]; print("$1", round(d["value"]), round(d["roofline"]["frac"],3), "total", round(p["total"],2), "update", round(p["update"],2), "panel", round(p["dbg"][1],2), "sweep wall", round(p["dbg"][7],2), d["solve_stats"]["all_solved"], d["solve_stats"].get("solution_sha256_16"))
PY
}
run default ""
for L in "$@"; do run $(basename $L) "--lib $L"; done
run default ""
