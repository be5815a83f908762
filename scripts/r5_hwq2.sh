#!/bin/bash
# 4 hardware queues: which of the step's 5 streams should exist?  (a fused early update off = 4 streams; caption stream off = 4 streams)
cd "$GRAFT_REPO_ROOT"
A="--steps 20 --warmup 3 --roofline-steps 0 --no-cpu-baseline --no-reference-precision --no-parity"
run() { timeout -k 10 300 python bench.py $A 2> gpurun_out/r5_hwq.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', o['value'], o['ms_per_step'])" || exit 1; }
for i in 1 2; do
run "default (5 streams)"
LECCR_CAP_STREAM=0 run "caption stream off (4 streams)"
LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0 run "text + caption streams off (3 streams)"
LECCR_BENCH_NO_FUSION=1 run "early update not fused (4 streams)"
done
