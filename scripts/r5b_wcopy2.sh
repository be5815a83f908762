#!/bin/bash
# round 5, second session: weight-copy kernels alone + their time inside the profiled step (compare profiles/r05_bench_b256_kernel_stats.md: 2 x 482.9 us)
O=gpurun_out/wcopy; mkdir -p $O
python3 scripts/wcopy_bench.py > $O/alone.txt 2>&1; cat $O/alone.txt
bash scripts/profile_bench.sh wcopy
f=$(find gpurun_out/prof_wcopy -name "*kernel_stats.csv" | head -1)
grep -i "transpose_cast_batched\|adamw" $f | cut -c1-200
