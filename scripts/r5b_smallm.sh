#!/bin/bash
O=gpurun_out/smallm; mkdir -p $O
for v in 0 2 1; do
SMALL_M=3462 VARIANT=$v LIBCMP=1 REPS=20 python3 scripts/gemm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gemm_small_m.txt
done
