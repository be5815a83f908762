#!/bin/bash
# fp8 GEMM k-step schedules (profiles/r04_fp8_kstep_schedules.txt): "burst" = the k-step issues its fragment reads / DMA pieces ahead of its MFMAs.
#   TN (gemm256t4_fp8_kernel): compile-time, bit ks of T4_FP8_SCHED.  Build the four variants first (here, no GPU needed):
#       for v in 0 1 2 3; do bash scripts/build_variant.sh t4s$v "-DT4_FP8_SCHED=$v" gemm256t4; done
#   NT (gemm256w4 fp8): schedules 0 (both interleaved) and 2 (second k-step bursts) are in the shipped library, LECCR_FP8_BURST1=0 / 1 forces one
#       (the sweep of all four was run from a build that still had the choice as a macro, round 4).
set -e
mkdir -p gpurun_out/sched
: > gpurun_out/sched/tn.txt; : > gpurun_out/sched/nt.txt
for rep in 1 2; do
for v in 0 1 2 3; do
  [ -f gpurun_variants/libleccr_t4s$v.so ] && LECCR_LIB=$PWD/gpurun_variants/libleccr_t4s$v.so REPS=10 python scripts/gemm_fp8_dw_bench.py 2>/dev/null | awk -v v=$v -v r=$rep '{print "TN sched " v " run " r ": " $0}' | cut -c1-200 >> gpurun_out/sched/tn.txt
done
for m in 0 1; do
  LECCR_FP8_BURST1=$m REPS=10 python scripts/gemm_fp8_bench.py 2>/dev/null | awk -v v=$m -v r=$rep '{print "NT burst1 " v " run " r ": " $0}' | cut -c1-170 >> gpurun_out/sched/nt.txt
done
done
cat gpurun_out/sched/tn.txt gpurun_out/sched/nt.txt
