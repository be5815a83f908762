#!/bin/bash
# LayerNorm pass writes the split copy (fp32 path): kernel test, fp32 model tests, then the fp32 headline with / without it
set -e
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "split_copy_of_its_output or layernorm" > gpurun_out/ln_s3_tests.log 2>&1
python -m pytest tests/test_model_gpu.py -q -x >> gpurun_out/ln_s3_tests.log 2>&1
tail -3 gpurun_out/ln_s3_tests.log
python bench.py --dtype fp32 --steps 8 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/ln_s3_on.json 2> gpurun_out/ln_s3_on.err
LECCR_X3_LN=0 python bench.py --dtype fp32 --steps 8 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/ln_s3_off.json 2> gpurun_out/ln_s3_off.err
python bench.py --dtype fp32 --steps 8 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/ln_s3_on2.json 2> gpurun_out/ln_s3_on2.err
grep -h -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ln_s3_on.json gpurun_out/ln_s3_off.json gpurun_out/ln_s3_on2.json
