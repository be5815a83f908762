#!/bin/bash
# closing lines on one box after the T4_SCHED fix: suite, smoke, bf16 headline (with cpu_baseline), fp32, B/32 fp32, config E bf16 / fp8, config B fp8
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
python -m pytest tests -m gpu -q > $O/check9_tests.log 2>&1; tail -2 $O/check9_tests.log
python __graft_entry__.py smoke 2>&1 | tail -1
run() { tag=$1; shift; python3 $R/bench.py "$@" > $O/r4g_$tag.json 2> $O/r4g_$tag.err || echo "$tag FAILED"; python3 - <<PY
import json
try:
    d = json.loads([l for l in open('$O/r4g_$tag.json') if l.startswith('{')][-1]); r = d['roofline']
    print('$tag', d['value'], d['unit'], d['ms_per_step'], 'roofline', r['achieved'], r['peak'], r['frac'], 'gemm', r.get('gemm_ms_per_step'), 'attn', r.get('attn_fwd_ms_per_step'), r.get('attn_bwd_ms_per_step'), r.get('attn_fwd_tflops'), r.get('attn_bwd_tflops'), 'hbm', d['config'].get('peak_hbm_gb'), (d.get('parity') or {}).get('max_abs_logits_err'), (d.get('parity') or {}).get('gt_ranks_equal_to_oracle_encoders'), (d.get('cpu_baseline') or {}).get('value'), flush=True)
except Exception as e:
    print('$tag', 'no json', e, flush=True)
PY
}
run b16_bf16
run b16_fp32 --dtype fp32 --no-cpu-baseline
run b32_fp32 --clip ViT-B/32 --dtype fp32 --no-cpu-baseline
run E_bf16 --clip ViT-L/14 --text xlmr-large --batch 256 --dtype bf16 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 1
run E_fp8 --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 1
run B_fp8 --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --roofline-steps 1
