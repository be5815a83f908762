#!/bin/bash
python -m pytest tests -m gpu -q 2>&1 | tail -4
for conf in "" "expandable_segments:True"; do PYTORCH_HIP_ALLOC_CONF=$conf python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);c=d['config'];print('alloc conf [$conf]',d['value'],d['ms_per_step'],c['peak_hbm_gb'],c['peak_hbm_reserved_gb'],c['alloc_retries'])"; done
