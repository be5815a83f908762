#!/bin/bash
# kernel stats of the fp32 (reference-precision) step
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fp32 -o fp32 -- python3 $R/bench.py --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --roofline-steps 0 > $R/gpurun_out/prof_fp32.log 2>&1
tail -1 $R/gpurun_out/prof_fp32.log | cut -c1-300
