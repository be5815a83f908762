#!/bin/bash
python -m pytest tests -m gpu -q 2>&1 | tail -5
python bench.py --mode eval --steps 3 --warmup 1 > gpurun_out/r4l_eval_bf16_yaml.json 2> gpurun_out/r4l_eval_bf16_yaml.err
python bench.py --mode eval --steps 2 --warmup 1 --eval-no-regroup > gpurun_out/r4l_eval_bf16_yaml_asis.json 2>> gpurun_out/r4l_eval_bf16_yaml.err
python bench.py --mode eval --steps 2 --warmup 1 --dtype fp32 > gpurun_out/r4l_eval_fp32_yaml.json 2>> gpurun_out/r4l_eval_bf16_yaml.err
for t in eval_bf16_yaml eval_bf16_yaml_asis eval_fp32_yaml; do python -c "
import json;d=json.load(open('gpurun_out/r4l_$t.json'));print('$t',d['value'],d['texts_per_s'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'],d['roofline']['launches_per_pass'],d['parity'])"; done
