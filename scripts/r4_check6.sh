#!/bin/bash
for dt in fp8 bf16; do python bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype $dt --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 1 > gpurun_out/r4l_E_$dt.json 2> gpurun_out/r4l_E_$dt.err; python -c "
import json;d=json.loads(open('gpurun_out/r4l_E_$dt.json').read().strip().splitlines()[-1]);c=d['config'];print('E $dt',d['value'],d['ms_per_step'],d['ms_per_step_median'],d['roofline']['frac'],c['peak_hbm_gb'],c['peak_hbm_reserved_gb'],c['alloc_retries'],d['parity']['max_abs_logits_err'],d['parity']['ok'])"; done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('B bf16',d['value'],d['ms_per_step'],d['config']['peak_hbm_gb'],d['config']['peak_hbm_reserved_gb'])"
python -m pytest tests/test_model_gpu.py -q -k "fp8 or side_streams" 2>&1 | tail -2
