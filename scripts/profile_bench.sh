#!/bin/bash
# profile_bench.sh TAG [ENV=VAL ...] : rocprofv3 kernel stats of the default bench command -> gpurun_out/prof_TAG/
tag=$1; shift
for kv in "$@"; do export "$kv"; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o $tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-reference-precision --roofline-steps 0 > $R/gpurun_out/prof_$tag.log 2>&1
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-200
find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1
