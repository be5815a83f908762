#!/bin/bash
# run_exp.sh "1 2 4 ..." : gemm_bench on the diagnostic builds gpurun_variants/libleccr_w4e<N>.so of the 4-wave kernel
# (scripts/build_variant.sh w4e<N> -DW4_EXP=<N>; W4_EXP bits: 1 no vmcnt wait, 2 no DMA, 4 no barrier, 8 no fragment reads)
for e in ${1:-1 2 4 5 10}; do
  echo "== exp $e"
  LECCR_LIB=gpurun_variants/libleccr_w4e$e.so VARIANT=3 ONLY="fc1,fc2,sq 4096,sq 8192" timeout -k 10 100 python scripts/gemm_bench.py 2>&1 | grep TFLOP
done
