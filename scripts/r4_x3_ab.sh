#!/bin/bash
# fp32 step A/B of one ops switch: r4_x3_ab.sh LECCR_X3_SHARE   (or LECCR_X3_ATTN, LECCR_X3_LN, LECCR_X3_FUSE)
v=${1:-LECCR_X3_SHARE}
mkdir -p gpurun_out
for m in 1 0 1 0; do
env $v=$m python bench.py --dtype fp32 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --roofline-steps 0 > gpurun_out/x3ab_$m.json 2> gpurun_out/x3ab_$m.err
python3 - <<PY
import json
d = json.loads([l for l in open('gpurun_out/x3ab_$m.json') if l.startswith('{')][-1]); print('$v', $m, d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'), flush=True)
PY
done
