#!/bin/bash
# round 5: gemm256n4 v2 (128-byte rows, just-in-time fragments, parked output tile): parity, yardstick, schedule variants
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm256_n4" > gpurun_out/r5_n4_tests.log 2>&1 || { tail -40 gpurun_out/r5_n4_tests.log; exit 1; }
tail -1 gpurun_out/r5_n4_tests.log
S="qkv,out,fc1,fc2,sq 4096,sq 8192"
{
echo "== 8-wave kernel (variant 2) + vendor"; VARIANT=2 LIBCMP=1 REPS=10 ONLY="$S,bert qkv,bert fc1,bert fc2" python scripts/gemm_bench.py
echo "== n4 v2"; VARIANT=5 REPS=10 ONLY="$S,bert qkv,bert fc1,bert fc2" python scripts/gemm_bench.py
for v in $N4_VARIANTS; do
  echo "== n4 variant $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
done
echo "== n4 v2 again"; VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_n4_d.txt
