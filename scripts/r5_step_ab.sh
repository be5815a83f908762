#!/bin/bash
# round 5: the whole step with the NT products on the 4-wave n4 kernel (LECCR_GEMM_N4=1) or the 8-wave kernel (0), A/B on one box
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in 0 1 0 1; do
  echo "== step LECCR_GEMM_N4=$v"; LECCR_GEMM_N4=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], d.get('parity'))"
done
} 2>&1 | tee gpurun_out/r5_step_ab.txt
