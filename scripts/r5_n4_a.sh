#!/bin/bash
# round 5: first run of the 4-wave 16x16x32 NT kernel (gemm256n4.hip): parity, then the yardstick shapes against the 8-wave kernel and
# the vendor library, then schedule / diagnostic variants built by scripts/build_variant.sh
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm256_n4" > gpurun_out/r5_n4_tests.log 2>&1 || { tail -30 gpurun_out/r5_n4_tests.log; exit 1; }
tail -3 gpurun_out/r5_n4_tests.log
NT="qkv,out,fc1,fc2,bert out,bert qkv,bert fc1,bert fc2,cap qkv,sq 4096,sq 8192"
{
echo "== 8-wave kernel (variant 2) + vendor"; VARIANT=2 LIBCMP=1 REPS=10 ONLY="$NT" python scripts/gemm_bench.py
echo "== n4 base (variant 5)"; VARIANT=5 REPS=10 ONLY="$NT" python scripts/gemm_bench.py
for v in $N4_VARIANTS; do
  echo "== n4 variant $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so VARIANT=5 REPS=10 ONLY="qkv,out,fc1,fc2,sq 4096,sq 8192" python scripts/gemm_bench.py
done
echo "== n4 base again"; VARIANT=5 REPS=10 ONLY="qkv,out,fc1,fc2,sq 4096,sq 8192" python scripts/gemm_bench.py
echo "== n4 colgroup 4"; LECCR_N4_COLGROUP=4 VARIANT=5 REPS=10 ONLY="qkv,fc1,sq 8192" python scripts/gemm_bench.py
echo "== n4 colgroup 8"; LECCR_N4_COLGROUP=8 VARIANT=5 REPS=10 ONLY="qkv,fc1,sq 8192" python scripts/gemm_bench.py
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_n4_a.txt
