#!/bin/bash
# bf16 weight-gradient kernel: k-step schedules (bit ks = burst), step shapes of config B
set -e
mkdir -p gpurun_out/sched
for rep in 1 2; do
for v in 0 1 8 9 15; do
  LECCR_LIB=$PWD/gpurun_variants/libleccr_t4b$v.so REPS=10 ONLY="dW qkv,dW out,dW fc1,dW fc2,bert dW fc1" python scripts/gemm_bench.py 2>/dev/null | grep dW | awk -v v=$v -v r=$rep '{print "bf16 TN sched " v " run " r ": " $0}' | cut -c1-130 >> gpurun_out/sched/tn_bf16.txt
done
done
cat gpurun_out/sched/tn_bf16.txt
