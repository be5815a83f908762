#!/bin/bash
for a in "--dtype fp32" "--dtype fp32 --clip ViT-B/32" ""; do python bench.py $a --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=d['roofline'];print('[$a]',d['value'],d['ms_per_step'],d['ms_per_step_median'],r['achieved'],r['frac'],r['gemm_ms_per_step'],d['config']['peak_hbm_gb'],d['parity']['max_abs_logits_err'],d['parity']['gt_ranks_equal_to_oracle_encoders'])"; done
python -m pytest tests/test_model_gpu.py -q -k "side_streams or fused_into or resume" 2>&1 | tail -2
