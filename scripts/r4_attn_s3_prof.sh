#!/bin/bash
# kernel trace of the fp32 step with / without the attention-written operand copies
R=$PWD
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  LECCR_X3_ATTN=$m rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_s3_$m -o s3_$m -- python3 $R/bench.py --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --roofline-steps 0 > $R/gpurun_out/prof_s3_$m.log 2>&1
done
cd $R
for m in 1 0; do
  python3 scripts/steady_stats.py $(find gpurun_out/prof_s3_$m -name "*kernel_trace.csv" | head -1) "fp32 step, LECCR_X3_ATTN=$m" "scripts/r4_attn_s3_prof.sh" > gpurun_out/prof_s3_$m.md
  grep -i "x3a\|split3\|colsum\|steady-state" gpurun_out/prof_s3_$m.md | cut -c1-140
done
