#!/bin/bash
# r5_lines.sh : the round's closing bench lines on one box (final build)
cd "$(dirname "$0")/.."
O=gpurun_out
: > $O/r5_lines.jsonl
line() { tag=$1; shift; echo "== $tag: $*" >&2; "$@" > $O/line_$tag.json 2> $O/line_$tag.err; rc=$?; tail -1 $O/line_$tag.json >> $O/r5_lines.jsonl; python3 - "$tag" $rc <<'PY'
import json, sys
tag, rc = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f'gpurun_out/line_{tag}.json').read().strip().splitlines()[-1])
    rp = d.get('reference_precision') or {}
    par = d.get('parity') or {}
    print(tag, 'rc', rc, d.get('value'), d.get('unit'), d.get('ms_per_step'), 'frac', (d.get('roofline') or {}).get('frac'), 'gemm_ms', (d.get('roofline') or {}).get('gemm_ms_per_step'),
          'logits_err', par.get('max_abs_logits_err'), 'ranks', par.get('gt_ranks_equal_to_oracle_encoders'), 'ref_prec', rp.get('pairs_s'), rp.get('max_abs_logits_err'), rp.get('gt_ranks'), flush=True)
except Exception as e:
    print(tag, 'rc', rc, 'no line', e, flush=True)
PY
}
line default python3 bench.py
line fp8_B python3 bench.py --dtype fp8 --no-cpu-baseline --no-reference-precision
line E_bf16 python3 bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype bf16 --parity-rows 4 --no-cpu-baseline --no-reference-precision --steps 6 --warmup 2
line E_fp8 python3 bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --parity-rows 4 --no-cpu-baseline --no-reference-precision --steps 6 --warmup 2
line eval python3 bench.py --mode eval
line b32_fp32 python3 bench.py --clip ViT-B/32 --dtype fp32 --no-cpu-baseline --no-reference-precision
