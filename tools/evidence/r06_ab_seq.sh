mkdir -p gpurun_out/r06_ab1
for v in "-2" "1" "-2" "1"; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu --no-mpc --sequential-rank-sums $v > gpurun_out/r06_ab1/seq_$v.$RANDOM.json 2>> gpurun_out/r06_ab1/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_ab1/seq_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); p=d["solve_stats"]["phase_ms_per_qp"]
        print(f, round(d["value"]), round(d["roofline"]["frac"],4), {k:round(v,2) for k,v in p.items() if k!="dbg"}, d["solve_stats"]["solution_sha256_16"], d["solve_stats"]["iter_mean"])
    except Exception as e: print(f,"FAILED",e)
PY
