O=gpurun_out/r06_stage3; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "work_queue or coop_update_sweep or downdate_into or linear_programmes" > $O/pytest_sel.log 2>&1; tail -8 $O/pytest_sel.log
timeout 300 python tools/evidence/fuzz_parity.py 702 300 hip 2 70 lp=1 > $O/lp_small_702.log 2>&1; tail -1 $O/lp_small_702.log
