#!/bin/bash
# r4_lines_b.sh : part B (config E geometry bf16 / fp8 with a 4-row parity probe, the two video workloads)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
run() { tag=$1; shift; python3 $R/bench.py "$@" > $O/r4l_$tag.json 2> $O/r4l_$tag.err || echo "$tag FAILED"; python3 - <<PY
import json
try:
    d = json.loads(open('$O/r4l_$tag.json').read().strip().splitlines()[-1]); r = d['roofline']
    print('$tag', d['value'], d['unit'], d['ms_per_step'], 'roofline', r['achieved'], r['peak'], r['frac'], 'hbm', d['config'].get('peak_hbm_gb'), d['config'].get('peak_hbm_reserved_gb'), (d.get('parity') or {}).get('max_abs_logits_err'), (d.get('parity') or {}).get('gt_ranks_equal_to_oracle_encoders'), flush=True)
except Exception as e:
    print('$tag', 'no json', e, flush=True)
PY
}
run E_bf16 --clip ViT-L/14 --text xlmr-large --batch 256 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 1
run E_fp8 --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 1
run video --model video --no-cpu-baseline
run video_vit --model video_vit --no-cpu-baseline
