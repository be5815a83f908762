#!/bin/bash
# round 5: the activation derivative saved by the forward epilogue (LECCR_ACT_DERIV=1, default) against the pre-activation (0): kernel tests, then lines
cd "$GRAFT_REPO_ROOT"
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "gemm" > gpurun_out/r5_actd.log 2>&1; rc=$?; tail -3 gpurun_out/r5_actd.log
[ $rc -ne 0 ] && exit $rc
run() { timeout -k 10 300 python bench.py $A 2> gpurun_out/r5_actd.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=o['roofline']; print('$1', o['value'], o['ms_per_step'], r['frac'], r['gemm_ms_per_step'])" || exit 1; }
A="--steps 12 --warmup 3 --no-parity --no-cpu-baseline --no-reference-precision"
for v in 0 1 0 1; do LECCR_ACT_DERIV=$v run "config B bf16 LECCR_ACT_DERIV=$v"; done
A="--clip ViT-L/14 --text xlmr-large --batch 256 --no-parity --no-cpu-baseline --no-reference-precision --steps 6 --warmup 2 --roofline-steps 2 --dtype fp8"
for v in 0 1 0 1; do LECCR_ACT_DERIV=$v run "config E fp8 LECCR_ACT_DERIV=$v"; done
A="--clip ViT-L/14 --text xlmr-large --batch 256 --no-parity --no-cpu-baseline --no-reference-precision --steps 6 --warmup 2 --roofline-steps 2 --dtype bf16"
for v in 0 1; do LECCR_ACT_DERIV=$v run "config E bf16 LECCR_ACT_DERIV=$v"; done
A="--steps 6 --warmup 2 --no-parity --no-cpu-baseline --no-reference-precision --dtype fp32"
for v in 0 1; do LECCR_ACT_DERIV=$v run "config B fp32 LECCR_ACT_DERIV=$v"; done
