#!/bin/bash
# round 5, second session: the fp32 path's LayerNorm backward passes write the G-form copy of the gradient the next products read (LECCR_X3_LN_BWD)
O=gpurun_out/lnbwd3; mkdir -p $O
python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "layernorm" > $O/tests_kernel.log 2>&1; echo "kernel tests rc $?"; tail -3 $O/tests_kernel.log
python3 -m pytest tests/test_model_gpu.py tests/test_full_geometry_gpu.py -x -q -m gpu -k "fp32 or f32 or float32" > $O/tests_model.log 2>&1; echo "model tests rc $?"; tail -2 $O/tests_model.log
B="--dtype fp32 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 16"
for m in 0 1 0 1; do
LECCR_X3_LN_BWD=$m python3 bench.py $B > $O/line_$m.json 2> $O/line_$m.err
python3 - $m <<'P'
import json, sys
m = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/lnbwd3/line_%s.json' % m) if l.startswith('{')][-1])
r, p = d['roofline'], d.get('parity', {})
print('LECCR_X3_LN_BWD=' + m, d['value'], d['ms_per_step'], 'logits err', p.get('max_abs_logits_err'), 'ranks', p.get('gt_ranks_equal_to_oracle_encoders'), 'loss err', p.get('max_abs_loss_err'), d['config'].get('peak_hbm_gb'), flush=True)
P
done
