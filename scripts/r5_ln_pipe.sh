#!/bin/bash
# round 5: software-pipelined LayerNorm backward (LECCR_LN_BWD_PIPE=1, default) against the plain kernel (0): parity tests, the kernel alone, the step
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "layernorm" > gpurun_out/r5_ln_tests.log 2>&1 || { tail -40 gpurun_out/r5_ln_tests.log; exit 1; }
tail -1 gpurun_out/r5_ln_tests.log
{
for v in 0 1 0 1; do echo "== LECCR_LN_BWD_PIPE=$v"; LECCR_LN_BWD_PIPE=$v python scripts/ln_bench.py; done
for v in 0 1 0 1; do
  echo "== step LECCR_LN_BWD_PIPE=$v"; LECCR_LN_BWD_PIPE=$v python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-reference-precision 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], [(h['kernel'], h['ms_per_step'], h['achieved_tb_s']) for h in d['hbm_kernels'][:2]])"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5_ln_pipe.txt
