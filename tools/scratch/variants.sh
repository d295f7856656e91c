#!/bin/bash
# same-box comparison of build variants: tools/scratch/variants.sh "name:extra bench args" ...   (name = suffix of qpalm_amd/lib/libqpalm_gfx950_<name>.so, cur = the shipped build)
mkdir -p gpurun_out
for v in "$@"; do
  name=${v%%:*}; args=${v#*:}
  lib=$PWD/qpalm_amd/lib/libqpalm_gfx950_$name.so
  [ "$name" = "cur" ] && lib=$PWD/qpalm_amd/lib/libqpalm_gfx950.so
  timeout 600 python bench.py --lib $lib --steps 2 --warmup 1 --no-cpu $args > gpurun_out/var_${name}_$(echo $args | tr -d ' -').json 2>> gpurun_out/var.err
done
