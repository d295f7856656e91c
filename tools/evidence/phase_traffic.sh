#!/bin/bash
# Per-phase HBM traffic of the solver's kernels (run on the GPU box):  bash tools/evidence/phase_traffic.sh [outdir-name]
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (counters only), then the summary.
R=${1:-r04/phase_traffic}
shift
ARGS="$@"   # passed on to tools/evidence/phase_traffic.py (e.g. --opt ld_align=8)
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/tools/evidence/phase_traffic.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/tools/evidence/phase_traffic.py $ARGS > $OUT/write.json 2> $OUT/write.err
python3 $REPO/tools/evidence/phase_traffic.py --summarise $OUT > $OUT/phase_traffic.json
find $OUT/pmc_fetch $OUT/pmc_write -name "*counter_collection*.csv" | head -4
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/phase_traffic.json
