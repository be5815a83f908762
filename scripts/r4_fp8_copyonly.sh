#!/bin/bash
# fp8 step: producers that write only the fp8 copy where nothing reads the bf16 tensor (LECCR_FP8_ATTN_COPY_ONLY: attention backward; LECCR_FP8_LN_COPY_ONLY: the blocks' LayerNorm passes - the one A/B-ed below): tests, then config B / E fp8 lines
set -e
O=gpurun_out/fp8step; mkdir -p $O
python -m pytest tests -q -x -m gpu -k "fp8" > $O/copyonly_tests.log 2>&1 || { tail -40 $O/copyonly_tests.log; exit 1; }
tail -1 $O/copyonly_tests.log
line() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -5 $O/$tag.err; }; python3 - <<PY
import json
d = json.loads([l for l in open('$O/$tag.json') if l.startswith('{')][-1]); print('$tag', d['value'], d['ms_per_step'], d['config'].get('peak_hbm_gb'), flush=True)
PY
}
B="--dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --roofline-steps 0"
E="--clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0"
for m in 1 0 1 0; do line B_lnonly_$m env LECCR_FP8_LN_COPY_ONLY=$m python3 bench.py $B; done
for m in 1 0; do line E_lnonly_$m env LECCR_FP8_LN_COPY_ONLY=$m python3 bench.py $E; done
