O=gpurun_out/r06/fuzz_final; mkdir -p $O
F=tools/evidence/fuzz_parity.py
for mode in 1 0; do
  timeout 900 python $F 721 200 hip 70 300 q_scale=1e-10 factorization_method=1 sequential_rank_sums=$mode > $O/tiny_q_721_rank_sums_$mode.log 2>&1; tail -1 $O/tiny_q_721_rank_sums_$mode.log
done
