#!/bin/bash
# same-box A/B of library builds (run on the GPU box): tools/scratch/ab_r04.sh tag name [name ...]   (name = cur or the suffix of libqpalm_gfx950_<name>.so)
# per build: two default bench runs (interleaved) and the per-phase traffic passes -> gpurun_out/<tag>/
tag=$1; shift
REPO=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $REPO/gpurun_out/$tag
for rep in 1 2; do
  for name in "$@"; do
    lib=$REPO/qpalm_amd/lib/libqpalm_gfx950_$name.so; [ "$name" = "cur" ] && lib=$REPO/qpalm_amd/lib/libqpalm_gfx950.so
    (cd $REPO && timeout 900 python bench.py --no-cpu --lib $lib > gpurun_out/$tag/bench_${name}_$rep.json 2>> gpurun_out/$tag/bench.err)
  done
done
if [ -z "$AB_NO_TRAFFIC" ]; then
for name in "$@"; do
  lib=$REPO/qpalm_amd/lib/libqpalm_gfx950_$name.so; [ "$name" = "cur" ] && lib=$REPO/qpalm_amd/lib/libqpalm_gfx950.so
  (cd $REPO && bash tools/evidence/phase_traffic.sh $tag/pt_$name --lib $lib > gpurun_out/$tag/pt_$name.log 2>&1)
done
fi
cd $REPO && python3 tools/scratch/ab_r04_show.py gpurun_out/$tag "$@"
