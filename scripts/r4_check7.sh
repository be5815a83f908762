#!/bin/bash
python -m pytest tests -m gpu -q 2>&1 | tail -3
python bench.py --mode eval --steps 3 --warmup 1 > gpurun_out/r4l_eval_bf16_yaml.json 2>/dev/null
python bench.py --mode eval --steps 2 --warmup 1 --dtype fp32 > gpurun_out/r4l_eval_fp32_yaml.json 2>/dev/null
for t in eval_bf16_yaml eval_fp32_yaml; do python -c "
import json;d=json.load(open('gpurun_out/r4l_$t.json'));print('$t',d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['frac'],d['parity']['max_abs_score_err'],d['parity']['subsample_ranks_equal'],d['parity']['ok'])"; done
