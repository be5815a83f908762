#!/bin/bash
# round 5: non-temporal output stores in the 8-wave kernel - yardstick shapes and the whole step, A/B on one box
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
S="qkv,out,fc1,fc2,bert qkv,bert fc1,bert fc2,sq 4096,sq 8192"
{
for v in g256_plain g256_nt g256_plain g256_nt; do
  echo "== $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so LECCR_GEMM_N4=0 REPS=10 ONLY="$S" python scripts/gemm_bench.py
done
for v in g256_plain g256_nt g256_plain g256_nt; do
  echo "== step $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so LECCR_GEMM_N4=0 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
done
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_nt_ab.txt
