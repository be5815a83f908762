#!/bin/bash
# round 5, second session: bias gradients of the fp32 path summed inside the weight-gradient launches (leccr_gemm_params.asum_x3): kernel test, the fp32
# model tests, then the fp32 step A/B on one box (LECCR_X3_ASUM=0: colsum passes over the split copy; 1: c_fc / intermediate.dense; 2: every block Linear)
O=gpurun_out/x3asum; mkdir -p $O
python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "sums_the_bias_gradient_from_the_split_copy or tn_fused_bias or f32_weight_gradient" > $O/tests_kernel.log 2>&1; echo "kernel tests rc $?"; tail -3 $O/tests_kernel.log
for m in 1 2; do
LECCR_X3_ASUM=$m python3 -m pytest tests/test_model_gpu.py tests/test_full_geometry_gpu.py -x -q -m gpu -k "fp32 or f32 or float32" > $O/tests_model_$m.log 2>&1; echo "model tests (LECCR_X3_ASUM=$m) rc $?"; tail -2 $O/tests_model_$m.log
done
B="--dtype fp32 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 16"
for m in 0 1 2 0 1 2; do
LECCR_X3_ASUM=$m python3 bench.py $B > $O/line_$m.json 2> $O/line_$m.err
python3 - $m <<'P'
import json, sys
m = sys.argv[1]
d = json.loads([l for l in open('gpurun_out/x3asum/line_%s.json' % m) if l.startswith('{')][-1])
r, p = d['roofline'], d.get('parity', {})
print('LECCR_X3_ASUM=' + m, d['value'], d['ms_per_step'], 'split ms', r.get('split_ms_per_step'), 'logits err', p.get('max_abs_logits_err'), 'ranks', p.get('gt_ranks_equal_to_oracle_encoders'), 'loss err', p.get('max_abs_loss_err'), flush=True)
P
done
