#!/bin/bash
# fp8 GEMM kernels with both k-steps' fills between the MFMAs again (advance() hoisted), + asum in the fp8 weight-gradient kernel:
# kernel tests, then the fp8 GEMM benches with the previous build (gpurun_variants/libleccr_presink.so) and this one, same box
set -e
mkdir -p gpurun_out/sink
python -m pytest tests/test_kernels_gpu.py -q -x -k "fp8" > gpurun_out/sink/tests.log 2>&1 || { tail -30 gpurun_out/sink/tests.log; exit 1; }
tail -2 gpurun_out/sink/tests.log
LECCR_LIB=$PWD/gpurun_variants/libleccr_presink.so python scripts/gemm_fp8_bench.py > gpurun_out/sink/nt_old.txt 2>&1
python scripts/gemm_fp8_bench.py > gpurun_out/sink/nt_new.txt 2>&1
LECCR_LIB=$PWD/gpurun_variants/libleccr_presink.so python scripts/gemm_fp8_dw_bench.py > gpurun_out/sink/tn_old.txt 2>&1
python scripts/gemm_fp8_dw_bench.py > gpurun_out/sink/tn_new.txt 2>&1
paste -d'\n' gpurun_out/sink/nt_old.txt gpurun_out/sink/nt_new.txt | cut -c1-150
paste -d'\n' gpurun_out/sink/tn_old.txt gpurun_out/sink/tn_new.txt | cut -c1-170
