#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
S="qkv,out,fc1,fc2,sq 8192"
{
echo "== n4 v2"; VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
for v in $N4_VARIANTS; do
  echo "== n4 variant $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
done
echo "== 8-wave kernel (variant 2)"; VARIANT=2 REPS=10 ONLY="$S" python scripts/gemm_bench.py
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_n4_e.txt
