#!/bin/bash
# round 6, stage 2: LP campaigns with the non-finite-direction guard only; queue order A/B on the headline; the whole GPU suite
O=gpurun_out/r06_stage2; mkdir -p $O
timeout 600 python tools/evidence/fuzz_parity.py 701 200 hip 70 400 lp=1 factorization_method=1 > $O/lp_schur_701.log 2>&1; tail -1 $O/lp_schur_701.log
timeout 300 python tools/evidence/fuzz_parity.py 702 300 hip 2 70 lp=1 > $O/lp_small_702.log 2>&1; tail -1 $O/lp_small_702.log
for v in 1 0 1 0; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-mpc --opt queue_order=$v > $O/bench_queue_$v.$RANDOM.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_stage2/bench_queue_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); p=d["solve_stats"]["phase_ms_per_qp"]
        print(f.split("/")[-1], round(d["value"]), round(d["roofline"]["frac"],4), "kernel_ms", round(d["roofline"]["kernel_ms"],1), {k:round(v,2) for k,v in p.items() if k!="dbg"}, d["solve_stats"]["solution_sha256_16"], d["solve_stats"]["pivot_guard"])
    except Exception as e: print(f,"FAILED",e)
PY
timeout 2400 python -m pytest tests -q -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
