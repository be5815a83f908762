#!/bin/bash
# round 5: store policy of the 8-wave kernel's output tiles inside the step: 0 default, 1 non-temporal, 2 non-temporal for outputs >= 256 MB only
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for v in 0 1 2 0 1 2; do
  echo "== step LECCR_G256_NT=$v"; LECCR_G256_NT=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-reference-precision 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], [ (h['kernel'], h['ms_per_step']) for h in d['hbm_kernels'][:2]], d['roofline']['attn_fwd_ms_per_step'], d['roofline']['attn_bwd_ms_per_step'])"
done
} 2>&1 | tee gpurun_out/r5_nt_policy.txt
