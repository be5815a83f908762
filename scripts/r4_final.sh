#!/bin/bash
# r4_final.sh : closing checks of round 4 on one box: GPU suite, smoke(), the headline / fp32 / reference-default bench lines, fp32 kernel stats
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4
python __graft_entry__.py smoke 2>&1 | tail -1
run() { tag=$1; shift; python3 $R/bench.py "$@" > $O/r4f_$tag.json 2> $O/r4f_$tag.err || echo "$tag FAILED"; python3 - <<PY
import json
try:
    d = json.loads(open('$O/r4f_$tag.json').read().strip().splitlines()[-1]); r = d['roofline']
    print('$tag', d['value'], d['unit'], d['ms_per_step'], d.get('ms_per_step_median'), 'roofline', r['achieved'], r['peak'], r['frac'], 'gemm', r.get('gemm_ms_per_step'), 'attn', r.get('attn_fwd_ms_per_step'), r.get('attn_bwd_ms_per_step'), 'hbm', d['config'].get('peak_hbm_gb'), d['config'].get('peak_hbm_reserved_gb'), (d.get('parity') or {}).get('max_abs_logits_err'), (d.get('parity') or {}).get('gt_ranks_equal_to_oracle_encoders'), (d.get('cpu_baseline') or {}).get('value'), flush=True)
    print('   ', [(h['kernel'][:12], h['ms_per_step'], h['achieved_tb_s']) for h in d.get('hbm_kernels', [])])
except Exception as e:
    print('$tag', 'no json', e, flush=True)
PY
}
run b16_bf16
run b16_fp32 --dtype fp32 --no-cpu-baseline
run b32_fp32 --clip ViT-B/32 --dtype fp32 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r4f_fp32 -o r4f_fp32 -- python3 $R/bench.py --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --roofline-steps 0 > $O/prof_r4f_fp32.log 2>&1
find $O/prof_r4f_fp32 -name "*kernel_stats.csv" | head -1
