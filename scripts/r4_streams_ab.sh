#!/bin/bash
# side streams on / off, config E fp8 and config B bf16, one box
for cfg in "A=1" "LECCR_DW_STREAM=0" "LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0"; do env $cfg python bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('E fp8 $cfg',d['value'],d['ms_per_step'],d['ms_per_step_median'])"; done
for cfg in "A=1" "LECCR_DW_STREAM=0" "LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0"; do env $cfg python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('B bf16 $cfg',d['value'],d['ms_per_step'],d['ms_per_step_median'])"; done
