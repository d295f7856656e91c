#!/bin/bash
# round 6, stage 1: the per-column pivot guard + the non-finite-direction guard on the hardware: the tests that pin them, the LP campaigns with
# their logs, and the same-box price of the guard on the headline (default = guarded tree; 0 = unguarded tree; 1 = running pivot everywhere)
O=gpurun_out/r06_stage1; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "downdate_into or sigma_grown or linear_programmes or boundary_ldlupdate or near_zero_pivot" > $O/pytest_guard.log 2>&1; tail -12 $O/pytest_guard.log
timeout 600 python tools/evidence/fuzz_parity.py 701 200 hip 70 400 lp=1 factorization_method=1 > $O/lp_schur_701.log 2>&1; tail -1 $O/lp_schur_701.log
timeout 300 python tools/evidence/fuzz_parity.py 702 300 hip 2 70 lp=1 > $O/lp_small_702.log 2>&1; tail -1 $O/lp_small_702.log
for v in -1 0 -1 0 1; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-mpc --sequential-rank-sums $v > $O/bench_seq_$v.$RANDOM.json 2>> $O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_stage1/bench_seq_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); p=d["solve_stats"]["phase_ms_per_qp"]
        print(f.split("/")[-1], round(d["value"]), round(d["roofline"]["frac"],4), {k:round(v,2) for k,v in p.items() if k!="dbg"}, d["solve_stats"]["solution_sha256_16"], d["solve_stats"]["pivot_guard"])
    except Exception as e: print(f,"FAILED",e)
PY
