#!/bin/bash
O=gpurun_out/evalgraph; mkdir -p $O
for n in 1 2 3 4 6; do
LECCR_EVAL_GRAPH_INSTANCES=$n timeout -k 10 400 python3 bench.py --mode eval --eval-no-regroup --steps 3 --warmup 1 > $O/inst_$n.json 2> $O/inst_$n.err; echo "bench rc $?"
python3 - $n <<'P'
import json, sys
g = sys.argv[1]
try:
    d = json.loads([l for l in open('gpurun_out/evalgraph/inst_%s.json' % g) if l.startswith('{')][-1])
    print('instances=' + g, d['value'], 'images/s', d['ms_per_step'], 'ms/pass', d['parity']['max_abs_score_err'], d['config']['peak_hbm_gb'], flush=True)
except Exception as e:
    print('no line', e); print(open('gpurun_out/evalgraph/inst_%s.err' % g).read()[-3000:])
P
done
