#!/bin/bash
# fp8 GEMM k-step schedules: bit ks of the variant number = k-step ks issues its fragment reads / DMA pieces in one burst ahead of its MFMAs
set -e
mkdir -p gpurun_out/sched
for rep in 1 2; do
for v in 0 1 2 3; do
  LECCR_LIB=$PWD/gpurun_variants/libleccr_t4s$v.so REPS=10 python scripts/gemm_fp8_dw_bench.py 2>/dev/null | awk -v v=$v -v r=$rep '{print "TN sched " v " run " r ": " $0}' | cut -c1-200 >> gpurun_out/sched/tn.txt
  LECCR_LIB=$PWD/gpurun_variants/libleccr_w4s$v.so REPS=10 python scripts/gemm_fp8_bench.py 2>/dev/null | awk -v v=$v -v r=$rep '{print "NT sched " v " run " r ": " $0}' | cut -c1-170 >> gpurun_out/sched/nt.txt
done
done
sort -k6,7 -s gpurun_out/sched/tn.txt | awk '{print $1,$2,$3,$5, $6,$7, $(NF-4), $(NF-3)}'
sort -k6,7 -s gpurun_out/sched/nt.txt | awk '{print $1,$2,$3,$5, $6,$7, $18, $19}'
