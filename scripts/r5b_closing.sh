#!/bin/bash
# round 5, second session: closing evidence of the final build on one box - the default bench line (driver-shaped), kernel stats of the bf16 step (default
# schedule and single stream) and of the fp32 step, the eval lines
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out/closing; mkdir -p $O
cd $R
python3 bench.py > $O/line_default.json 2> $O/line_default.err; echo "default bench rc $?"
python3 - <<'P'
import json
d = json.loads([l for l in open('gpurun_out/closing/line_default.json') if l.startswith('{')][-1])
r = d['reference_precision']
print('bf16', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'attn', d['roofline']['attn_fwd_tflops'], d['roofline']['attn_bwd_tflops'], '| fp32 leg', r['pairs_s'], r['ms_per_step'], r['max_abs_logits_err'], r['gt_ranks'], '| cpu', d['cpu_baseline']['value'], flush=True)
P
bash scripts/profile_bench.sh r5b && bash scripts/profile_bench.sh r5b_ss LECCR_DW_STREAM=0 LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0 && bash scripts/r5b_fp32_trace.sh
python3 bench.py --mode eval --steps 4 --warmup 1 > $O/line_eval.json 2> $O/line_eval.err; echo "eval rc $?"
python3 bench.py --mode eval --eval-no-regroup --steps 3 --warmup 1 > $O/line_eval_as_yielded.json 2> $O/line_eval_as_yielded.err; echo "eval as yielded rc $?"
python3 bench.py --dtype fp32 --clip ViT-B/32 --steps 8 --warmup 2 --no-cpu-baseline > $O/line_b32_fp32.json 2> $O/line_b32_fp32.err; echo "B/32 fp32 rc $?"
for f in line_eval line_eval_as_yielded line_b32_fp32; do python3 -c "
import json
d = json.loads([l for l in open('gpurun_out/closing/$f.json') if l.startswith('{')][-1]); print('$f', d['value'], d['unit'], d['ms_per_step'], d.get('parity', {}).get('ok'))"; done
