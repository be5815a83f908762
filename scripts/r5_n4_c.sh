#!/bin/bash
# round 5: do the epilogue's stores drain under the next item's K loop when they are spread over its stages? (diagnostic builds, wrong results)
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
S="qkv,out,fc1,fc2"
{
echo "== n4 base (direct epilogue)"; VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
for v in $N4_VARIANTS; do
  echo "== n4 variant $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
done
echo "== 8-wave kernel (variant 2)"; VARIANT=2 REPS=10 ONLY="$S" python scripts/gemm_bench.py
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_n4_c.txt
