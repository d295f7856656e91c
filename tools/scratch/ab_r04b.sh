#!/bin/bash
# same-box A/B (run on the GPU box): tools/scratch/ab_r04b.sh tag "name:bench args" ...   name = cur or the suffix of libqpalm_gfx950_<name>.so
tag=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $REPO/gpurun_out/$tag
cd $REPO
for rep in 1 2; do
  k=0
  for v in "$@"; do
    name=${v%%:*}; args=${v#*:}; [ "$args" = "$v" ] && args=""
    lib=$REPO/qpalm_amd/lib/libqpalm_gfx950_$name.so; [ "$name" = "cur" ] && lib=$REPO/qpalm_amd/lib/libqpalm_gfx950.so
    timeout 400 python bench.py --no-cpu --no-mpc --lib $lib $args > gpurun_out/$tag/bench_${k}_$rep.json 2>> gpurun_out/$tag/bench.err
    k=$((k+1))
  done
done
python3 - "$tag" "$@" <<'PY'
import json, sys
tag = sys.argv[1]
for k, v in enumerate(sys.argv[2:]):
    vals, h, ph, pq, fr = [], None, None, None, None
    for rep in (1, 2):
        try:
            j = json.load(open("gpurun_out/%s/bench_%d_%d.json" % (tag, k, rep)))
            vals.append(round(j["value"])); h = j["solve_stats"]["solution_sha256_16"]; ph = j["solve_stats"]["phase_ms_per_qp"]; pq = j["solve_stats"]["per_qp_mean"]; fr = j["roofline"]["frac"]
        except Exception as e:
            vals.append("FAILED")
    line = "%-28s QP/s %s hash %s" % (v, vals, h)
    if ph:
        line += " frac %.3f ms/QP total %.1f update %.1f factor %.1f solve %.1f ls %.1f | sweeps %.1f | dbg0 %.1f panel %.1f trail %.1f dbg7 %.1f" % (
            fr, ph["total"], ph["update"], ph["factor"], ph["solve"], ph["linesearch"], pq["n_sweeps"], ph["dbg"][0], ph["dbg"][1], ph["dbg"][2], ph["dbg"][7])
    print(line)
PY
