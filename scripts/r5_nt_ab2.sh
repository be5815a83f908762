#!/bin/bash
set -e -o pipefail
cd "$(dirname "$0")/.."
S="qkv,out,fc1,fc2"
{
for v in g256_nt0 g256_nt1 g256_nt2 g256_nt0 g256_nt1 g256_nt2; do
  echo "== $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so REPS=10 ONLY="$S" python scripts/gemm_bench.py
done
for v in g256_nt0 g256_nt1 g256_nt2 g256_nt0 g256_nt1 g256_nt2; do
  echo "== step $v"; LECCR_LIB=gpurun_variants/libleccr_$v.so python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-reference-precision 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
done
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_nt_ab2.txt
