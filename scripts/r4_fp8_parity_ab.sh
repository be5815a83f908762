#!/bin/bash
# fp8 lines' parity probe with the bias sums inside the weight-gradient launch (default) and from the bf16 pass
O=gpurun_out/fp8step; mkdir -p $O
for m in 1 0 1 0; do
LECCR_FP8_BIAS_ASUM=$m python3 bench.py --dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --roofline-steps 0 > $O/par_B_$m.json 2> $O/par_B_$m.err
python3 -c "
import json
d = json.loads([l for l in open('$O/par_B_$m.json') if l.startswith('{')][-1]); p = d['parity']; print('B asum', $m, d['value'], p['max_abs_logits_err'], p['gt_ranks_equal_to_oracle_encoders'], p.get('max_abs_feature_err'), flush=True)"
done
for m in 1 0; do
LECCR_FP8_BIAS_ASUM=$m python3 bench.py --clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --parity-rows 4 --roofline-steps 0 > $O/par_E_$m.json 2> $O/par_E_$m.err
python3 -c "
import json
d = json.loads([l for l in open('$O/par_E_$m.json') if l.startswith('{')][-1]); p = d['parity']; print('E asum', $m, d['value'], p['max_abs_logits_err'], p['gt_ranks_equal_to_oracle_encoders'], p.get('max_abs_feature_err'), flush=True)"
done
