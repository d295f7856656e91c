#!/bin/bash
# Round-end evidence, run on the GPU box:  tools/round_artifacts.sh   (outputs under gpurun_out/r01/)
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out/r01
mkdir -p $OUT
cd $REPO
python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --batch 512 --no-cpu > $OUT/bench_b512.json 2>> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $REPO/bench.py --no-cpu > $OUT/kt_bench.json 2> $OUT/kt.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
cd $OUT
# keep only the small summaries (the traces are large)
find kt -name "*kernel_stats*.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<'PY'
import csv, glob, json
def total(d, counter):
    tot = 0.0; n = 0
    for f in glob.glob(d + "/**/*counter_collection*.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_solve" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                tot += float(r["Counter_Value"]); n += 1
    return tot, n
fe, nf = total("pmc_fetch", "FETCH_SIZE"); wr, nw = total("pmc_write", "WRITE_SIZE")
json.dump({"FETCH_SIZE_KB": fe, "rows_fetch": nf, "WRITE_SIZE_KB": wr, "rows_write": nw}, open("pmc_summary.json", "w"))
PY
rm -rf kt pmc_fetch pmc_write
ls -la $OUT
