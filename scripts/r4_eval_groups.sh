#!/bin/bash
for a in "--eval-batch 64 --eval-text-batch 256" "--eval-batch 128 --eval-text-batch 512" "--eval-batch 256 --eval-text-batch 1024"; do python bench.py --mode eval --steps 3 --warmup 1 $a 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$a',d['value'],d['ms_per_step'],d['roofline']['frac'],d['config']['peak_hbm_gb'])"; done
