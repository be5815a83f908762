#!/bin/bash
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): the step uses 5 streams without a process group and 7 with one
cd "$GRAFT_REPO_ROOT"
A="--steps 20 --warmup 3 --roofline-steps 0 --no-cpu-baseline --no-reference-precision --no-parity"
for q in 1 2 3 4 5 6 2 3 4; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py $A 2> gpurun_out/r5_hwq.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no process group, queues $q', o['value'], o['ms_per_step'])" || exit 1
done
