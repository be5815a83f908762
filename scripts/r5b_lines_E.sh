#!/bin/bash
# round 5, second session: the fp8 config-B line and the config-E lines of the final build on one box
cd "$(dirname "$0")/.."
O=gpurun_out/closingE; mkdir -p $O
line() { tag=$1; shift; timeout -k 10 500 "$@" > $O/line_$tag.json 2> $O/line_$tag.err; rc=$?; python3 - "$tag" $rc <<'PY'
import json, sys
tag, rc = sys.argv[1], sys.argv[2]
try:
    d = json.loads([l for l in open(f'gpurun_out/closingE/line_{tag}.json') if l.startswith('{')][-1])
    par = d.get('parity') or {}
    print(tag, 'rc', rc, d.get('value'), d.get('unit'), d.get('ms_per_step'), 'frac', (d.get('roofline') or {}).get('frac'), 'logits_err', par.get('max_abs_logits_err'), 'ranks', par.get('gt_ranks_equal_to_oracle_encoders'), 'hbm', d['config'].get('peak_hbm_gb'), flush=True)
except Exception as e:
    print(tag, 'rc', rc, 'no line', e, flush=True)
PY
}
line fp8_B python3 bench.py --dtype fp8 --no-cpu-baseline --no-reference-precision
line E_bf16 python3 bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype bf16 --parity-rows 4 --no-cpu-baseline --no-reference-precision --steps 6 --warmup 2
line E_fp8 python3 bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --parity-rows 4 --no-cpu-baseline --no-reference-precision --steps 6 --warmup 2
