#!/bin/bash
# round 5, second session: evaluation as the reference batches it (6 images / 32 texts per forward), launch by launch vs whole batches as replayed HIP graphs
O=gpurun_out/evalgraph; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_eval_gpu.py -x -q -m gpu -k "as_yielded or synthetic_loader" > $O/tests.log 2>&1; echo "tests rc $?"; tail -15 $O/tests.log
for g in 0 1; do
LECCR_EVAL_GRAPHS=$g timeout -k 10 400 python3 bench.py --mode eval --eval-no-regroup --steps 3 --warmup 1 > $O/line_$g.json 2> $O/line_$g.err; echo "bench rc $?"
python3 - $g <<'P'
import json, sys
g = sys.argv[1]
try:
    d = json.loads([l for l in open('gpurun_out/evalgraph/line_%s.json' % g) if l.startswith('{')][-1])
    print('LECCR_EVAL_GRAPHS=' + g, d['value'], 'images/s', d['ms_per_step'], 'ms/pass', d['parity'], flush=True)
except Exception as e:
    print('no line', e); print(open('gpurun_out/evalgraph/line_%s.err' % g).read()[-3000:])
P
done
