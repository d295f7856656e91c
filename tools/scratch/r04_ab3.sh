cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04/ab3
python -m pytest tests/test_sweep32.py tests/test_parity.py tests/test_full_size.py -x -q -m gpu > gpurun_out/r04/ab3/pytest_sweep32.log 2>&1; tail -3 gpurun_out/r04/ab3/pytest_sweep32.log
for rep in 1 2; do for sr in 32 16; do python bench.py --no-cpu --no-mpc --sweep-ranks $sr > gpurun_out/r04/ab3/bench_sr${sr}_$rep.json 2>> gpurun_out/r04/ab3/bench.err; done; done
python - <<'PY'
import json
for sr in (32,16):
    for rep in (1,2):
        try:
            j=json.load(open("gpurun_out/r04/ab3/bench_sr%d_%d.json"%(sr,rep)))
            ph=j["solve_stats"]["phase_ms_per_qp"]; pq=j["solve_stats"]["per_qp_mean"]
            print(sr,rep,round(j["value"]),j["solve_stats"]["solution_sha256_16"],"frac %.3f"%j["roofline"]["frac"],"ms/QP total %.1f update %.1f factor %.1f solve %.1f ls %.1f"%(ph["total"],ph["update"],ph["factor"],ph["solve"],ph["linesearch"]),"sweeps %.1f entries %.0f"%(pq["n_sweeps"],pq["sweep_entries"]), "dbg0 %.2f dbg1 %.2f dbg2 %.2f dbg7 %.2f"%(ph["dbg"][0],ph["dbg"][1],ph["dbg"][2],ph["dbg"][7]))
        except Exception as e: print(sr,rep,"FAILED",e)
PY
