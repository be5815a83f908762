#!/bin/bash
# round 5, second session: the batched weight-copy kernels on 64 x 64 tiles (tests, alone, in the step: bf16 headline and the fp32 leg)
O=gpurun_out/wcopy; mkdir -p $O
python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "optimizer or split3" > $O/tests.log 2>&1; echo "tests rc $?"; tail -2 $O/tests.log
python3 scripts/wcopy_bench.py > $O/alone.txt 2>&1; cat $O/alone.txt
python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --ref-steps 8 > $O/line.json 2> $O/line.err; echo "bench rc $?"
python3 - <<'P'
import json
d = json.loads([l for l in open('gpurun_out/wcopy/line.json') if l.startswith('{')][-1])
r = d['reference_precision']
print('bf16', d['value'], d['ms_per_step'], d['roofline']['frac'], '| fp32 leg', r['pairs_s'], r['ms_per_step'], r['roofline'].get('split_ms_per_step'), r['max_abs_logits_err'], r['gt_ranks'])
P
