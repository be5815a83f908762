#!/bin/bash
# bf16 weight-gradient kernel: k-step burst experiment (bit ks of T4_BURST; build: for v in 0 1 8 9; do bash scripts/build_variant.sh t4b$v "-DT4_BURST=$v" gemm256t4; done), step shapes of config B
set -e
mkdir -p gpurun_out/sched; : > gpurun_out/sched/tn_bf16.txt
for rep in 1 2; do
for v in 0 1 8 9; do
  LECCR_LIB=$PWD/gpurun_variants/libleccr_t4b$v.so REPS=10 ONLY="dW qkv,dW out,dW fc1,dW fc2,bert dW fc1" python scripts/gemm_bench.py 2>/dev/null | grep dW | awk -v v=$v -v r=$rep '{print "bf16 TN sched " v " run " r ": " $0}' | cut -c1-130 >> gpurun_out/sched/tn_bf16.txt
done
done
cat gpurun_out/sched/tn_bf16.txt
