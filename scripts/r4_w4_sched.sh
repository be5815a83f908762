#!/bin/bash
# bf16 4-wave NT kernel (leccr_set_gemm_variant(3)) with k-step schedules (bit ks = burst) against the 8-wave production kernel, step shapes
set -e
mkdir -p gpurun_out/sched; : > gpurun_out/sched/nt_bf16.txt
for rep in 1 2; do
  REPS=10 ONLY="qkv,out,fc1,fc2,sq 8192" python scripts/gemm_bench.py 2>/dev/null | grep "M=" | awk -v r=$rep '{print "8-wave run " r ": " $0}' | cut -c1-110 >> gpurun_out/sched/nt_bf16.txt
for v in 0 1 8 9 15; do
  LECCR_LIB=$PWD/gpurun_variants/libleccr_w4b$v.so VARIANT=3 REPS=10 ONLY="qkv,out,fc1,fc2,sq 8192" python scripts/gemm_bench.py 2>/dev/null | grep "M=" | awk -v v=$v -v r=$rep '{print "4-wave sched " v " run " r ": " $0}' | cut -c1-120 >> gpurun_out/sched/nt_bf16.txt
done
done
cat gpurun_out/sched/nt_bf16.txt
