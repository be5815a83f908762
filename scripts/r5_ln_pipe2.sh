#!/bin/bash
# LayerNorm backward pipelined: the EX outputs and the 1024-wide rows (config E); test, then config E lines A/B (LECCR_LN_BWD_PIPE=2: round-5 first step only)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "layernorm" > gpurun_out/r5_ln_pipe2.log 2>&1; rc=$?; tail -3 gpurun_out/r5_ln_pipe2.log
[ $rc -ne 0 ] && exit $rc
E="--clip ViT-L/14 --text xlmr-large --batch 256 --no-parity --no-cpu-baseline --no-reference-precision --steps 6 --warmup 2 --roofline-steps 2"
for dt in bf16 fp8; do for p in 2 1 2 1; do
  LECCR_LN_BWD_PIPE=$p timeout -k 10 300 python bench.py $E --dtype $dt 2> gpurun_out/r5_ln_pipe2.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config E $dt LECCR_LN_BWD_PIPE=$p', o['value'], o['ms_per_step'], [(h['kernel'], h['ms_per_step'], h['achieved_tb_s']) for h in o['hbm_kernels'] if 'layernorm' in h['kernel']])" || exit 1
done; done
for p in 2 1; do
  LECCR_LN_BWD_PIPE=$p timeout -k 10 300 python bench.py --steps 12 --warmup 3 --no-parity --no-cpu-baseline --no-reference-precision 2> gpurun_out/r5_ln_pipe2.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config B bf16 LECCR_LN_BWD_PIPE=$p', o['value'], o['ms_per_step'], [(h['kernel'], h['ms_per_step'], h['achieved_tb_s']) for h in o['hbm_kernels'] if 'layernorm' in h['kernel']])" || exit 1
done
