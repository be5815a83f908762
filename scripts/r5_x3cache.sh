#!/bin/bash
# round 5: the fp32 path with the weights' split copies made once per step (LECCR_X3_WCACHE=1, default) or at every use (0): tests, then A/B of the step
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_full_geometry_gpu.py tests/test_model_gpu.py -x -q -k "split3 or fp32 or f32" > gpurun_out/r5_x3cache_tests.log 2>&1 || { tail -40 gpurun_out/r5_x3cache_tests.log; exit 1; }
tail -2 gpurun_out/r5_x3cache_tests.log
{
for v in 0 1 0 1; do
  echo "== fp32 step LECCR_X3_WCACHE=$v"; LECCR_X3_WCACHE=$v python bench.py --dtype fp32 --steps 8 --warmup 3 --no-cpu-baseline --no-reference-precision 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'], [(h['kernel'][:8], h['ms_per_step']) for h in d['hbm_kernels'] if 'split' in h['kernel']], d['parity']['max_abs_logits_err'], d['parity']['gt_ranks_equal_to_oracle_encoders'], d['config']['peak_hbm_gb'])"
done
} 2>&1 | tee gpurun_out/r5_x3cache.txt
