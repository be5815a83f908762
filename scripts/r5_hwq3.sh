#!/bin/bash
# the early AdamW update on a stream of its own (shares the weight-gradient stream's hardware queue) or on the caption stream (idle in the backward pass)
cd "$GRAFT_REPO_ROOT"
A="--steps 20 --warmup 3 --roofline-steps 0 --no-cpu-baseline --no-reference-precision --no-parity"
run() { timeout -k 10 300 python bench.py $A 2> gpurun_out/r5_hwq.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', o['value'], o['ms_per_step'])" || exit 1; }
for i in 1 2 3; do
LECCR_UPDATE_STREAM=own run "update stream: own"
LECCR_UPDATE_STREAM=caption run "update stream: caption"
done
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_world_gpu.py tests/test_model_gpu.py tests/test_rccl_one_rank_gpu.py -q -m gpu -x > gpurun_out/r5_dist_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5_dist_tests.log; exit $rc
