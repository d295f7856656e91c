#!/bin/bash
# an experimental build of the HIP library next to the shipped one:
#   tools/evidence/build_variant.sh [--patch file.patch] [--ipra] <name> "<extra -D flags>"
# -> qpalm_amd/lib/libqpalm_gfx950_<name>.so (git-ignored, travels to the GPU box; run with bench.py --lib or tools/evidence/gpu_ab.sh)
# --patch: the sources are copied to a scratch directory and the patch (tools/variants/*.patch) is applied there.
set -e
cd "$(dirname "$0")/../.."
src=qpalm_amd/csrc
patch=""
ipra="-mllvm -enable-ipra=0"
while true; do
  case "$1" in
    --patch) patch=$(realpath "$2"); shift 2 ;;
    --ipra) ipra=""; shift ;;
    *) break ;;
  esac
done
name=$1; shift
if [ -n "$patch" ]; then
  tmp=$(mktemp -d /tmp/qpalm_variant_XXXX)
  mkdir -p $tmp/qpalm_amd $tmp/include
  cp -r qpalm_amd/csrc $tmp/qpalm_amd/
  cp include/*.h $tmp/include/
  (cd $tmp && patch -p1 -s < "$patch")
  src=$tmp/qpalm_amd/csrc
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off $ipra -Wno-unused-value \
  -Wno-constant-logical-operand $@ -o qpalm_amd/lib/libqpalm_gfx950_$name.so $src/qpalm_gfx950.hip
echo built qpalm_amd/lib/libqpalm_gfx950_$name.so
