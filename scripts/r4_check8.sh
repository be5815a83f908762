#!/bin/bash
# full GPU suite on this build, then the config-E lines (bf16 / fp8, 4-row parity probe) on the same box
set -e
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/check8_tests.log 2>&1 || { tail -40 gpurun_out/check8_tests.log; exit 1; }
tail -2 gpurun_out/check8_tests.log
R=$PWD; O=$R/gpurun_out
run() { tag=$1; shift; python3 $R/bench.py "$@" > $O/r4l_$tag.json 2> $O/r4l_$tag.err || echo "$tag FAILED"; python3 - <<PY
import json
try:
    d = json.loads([l for l in open('$O/r4l_$tag.json') if l.startswith('{')][-1]); r = d['roofline']
    print('$tag', d['value'], d['unit'], d['ms_per_step'], 'roofline', r['achieved'], r['peak'], r['frac'], 'hbm', d['config'].get('peak_hbm_gb'), (d.get('parity') or {}).get('max_abs_logits_err'), (d.get('parity') or {}).get('gt_ranks_equal_to_oracle_encoders'), flush=True)
except Exception as e:
    print('$tag', 'no json', e, flush=True)
PY
}
run E_bf16 --clip ViT-L/14 --text xlmr-large --batch 256 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 1
run E_fp8 --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 1
run B_fp8 --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --roofline-steps 1
