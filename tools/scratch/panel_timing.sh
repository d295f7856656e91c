#!/bin/bash
# diagnostic build (-DQP_PANEL_TIMING=1): where the panel wave of the update sweep spends its time
mkdir -p gpurun_out
python bench.py --steps 1 --warmup 0 --no-cpu --batch 1024 --lib qpalm_amd/lib/libqpalm_gfx950_ptime.so $@ > gpurun_out/ptime.json 2> gpurun_out/ptime.err
python - <<PY
import json; d=json.load(open("gpurun_out/ptime.json")); p=d["solve_stats"]["phase_ms_per_qp"]; g=p["dbg"]
print(round(d["value"]), "update", round(p["update"],2), "sweep wall", round(g[7],2), "staging", round(g[0],2), "panel busy", round(g[1],2),
      "= rows of block", round(g[8],2), "+ recurrence", round(g[9],2), "+ solve/write-back", round(g[10],2), "; barrier wait", round(g[11],2), "; last trailing wave", round(g[2],2))
print(d["solve_stats"]["per_qp_mean"])
PY
tail -2 gpurun_out/ptime.err
