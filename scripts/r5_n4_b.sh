#!/bin/bash
# round 5: the n4 kernel with the register-only epilogue; diagnostics that price the DMA pieces and the epilogue
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm256_n4" > gpurun_out/r5_n4_tests.log 2>&1 || { tail -30 gpurun_out/r5_n4_tests.log; exit 1; }
tail -1 gpurun_out/r5_n4_tests.log
S="qkv,out,fc1,fc2,sq 4096,sq 8192"
{
echo "== 8-wave kernel (variant 2)"; VARIANT=2 REPS=10 ONLY="$S" python scripts/gemm_bench.py
echo "== n4 base (direct epilogue)"; VARIANT=5 REPS=10 ONLY="$S,bert qkv,bert fc1,bert fc2" python scripts/gemm_bench.py
for v in $N4_VARIANTS; do
  echo "== n4 variant $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
done
echo "== n4 base again"; VARIANT=5 REPS=10 ONLY="$S" python scripts/gemm_bench.py
for d in 16 40 80; do echo "== n4 desync $d"; LECCR_N4_DESYNC=$d VARIANT=5 REPS=10 ONLY="qkv,out,fc1,fc2" python scripts/gemm_bench.py; done
echo "== n4 colgroup 4"; LECCR_N4_COLGROUP=4 VARIANT=5 REPS=10 ONLY="sq 4096,sq 8192" python scripts/gemm_bench.py
echo "== n4 colgroup 2"; LECCR_N4_COLGROUP=2 VARIANT=5 REPS=10 ONLY="sq 4096,sq 8192" python scripts/gemm_bench.py
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_n4_b.txt
