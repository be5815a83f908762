#!/bin/bash
# fp32 path: attention kernels write / read the operand copies of o and dqkv: kernel test, fp32 model tests, then the fp32 line with / without
set -e
mkdir -p gpurun_out
python -m pytest tests/test_kernels_gpu.py -q -x -k "operand_copies_of_its_outputs or attention_f32" > gpurun_out/attn_s3_tests.log 2>&1 || { tail -40 gpurun_out/attn_s3_tests.log; exit 1; }
tail -1 gpurun_out/attn_s3_tests.log
python -m pytest tests/test_full_geometry_gpu.py -q -x -k "fp32 or written_copies" >> gpurun_out/attn_s3_tests.log 2>&1 || { tail -40 gpurun_out/attn_s3_tests.log; exit 1; }
tail -1 gpurun_out/attn_s3_tests.log
python -m pytest tests/test_model_gpu.py -q -x >> gpurun_out/attn_s3_tests.log 2>&1 || { tail -40 gpurun_out/attn_s3_tests.log; exit 1; }
tail -1 gpurun_out/attn_s3_tests.log
for m in 1 0 1 0; do
LECCR_X3_ATTN=$m python bench.py --dtype fp32 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --roofline-steps 0 > gpurun_out/attn_s3_$m.json 2> gpurun_out/attn_s3_$m.err
python3 -c "
import json
d = json.loads([l for l in open('gpurun_out/attn_s3_$m.json') if l.startswith('{')][-1]); print('X3_ATTN', $m, d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'), flush=True)"
done
