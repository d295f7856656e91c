#!/bin/bash
# an experimental build of the HIP library next to the shipped one: tools/build_variant.sh <name> "<extra -D flags>"
# -> qpalm_amd/lib/libqpalm_gfx950_<name>.so (git-ignored, travels to the GPU box; run with bench.py --lib or tools/ab_multi.sh)
set -e
name=$1; shift
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -enable-ipra=0 -Wno-unused-value \
  -Wno-constant-logical-operand $@ -o qpalm_amd/lib/libqpalm_gfx950_$name.so qpalm_amd/csrc/qpalm_gfx950.hip
echo built qpalm_amd/lib/libqpalm_gfx950_$name.so
