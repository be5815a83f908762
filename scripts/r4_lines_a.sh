#!/bin/bash
# r4_lines_a.sh : round-4 bench lines on one box, part A (config B bf16 / fp32, the reference's B/32 tower, eval mode, vendor yardstick)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
run() { tag=$1; shift; python3 $R/bench.py "$@" > $O/r4l_$tag.json 2> $O/r4l_$tag.err || echo "$tag FAILED"; python3 - <<PY
import json
try:
    d = json.loads(open('$O/r4l_$tag.json').read().strip().splitlines()[-1]); r = d['roofline']
    print('$tag', d['value'], d['unit'], d['ms_per_step'], 'roofline', r['achieved'], r['peak'], r['frac'], 'hbm', d['config'].get('peak_hbm_gb'), (d.get('parity') or {}).get('max_abs_logits_err'), (d.get('parity') or {}).get('gt_ranks_equal_to_oracle_encoders'), flush=True)
except Exception as e:
    print('$tag', 'no json', e, flush=True)
PY
}
run b16_bf16
run b16_fp32 --dtype fp32 --no-cpu-baseline
run b32_bf16 --clip ViT-B/32 --no-cpu-baseline
run b32_fp32 --clip ViT-B/32 --dtype fp32 --no-cpu-baseline
run eval_bf16_yaml --mode eval --steps 2 --warmup 1
run eval_bf16_b64 --mode eval --steps 3 --warmup 1 --eval-batch 64 --eval-text-batch 256 --eval-texts 5000
run eval_fp32_b64 --mode eval --steps 2 --warmup 1 --eval-batch 64 --eval-text-batch 256 --eval-texts 5000 --dtype fp32
LIBCMP=1 python3 $R/scripts/gemm_bench.py > $O/r4l_gemm_vendor_yardstick.txt 2>&1; tail -20 $O/r4l_gemm_vendor_yardstick.txt
