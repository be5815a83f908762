#!/bin/bash
# config E fp8 step: K1 burst schedule of the fp8 NT kernel never / by K (default) / always, two rounds
set -e
R=$PWD; O=$R/gpurun_out/fp8step; mkdir -p $O
E="--clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0"
for rep in 1 2; do
for m in 0 -1 1; do
  LECCR_FP8_BURST1=$m python3 bench.py $E > $O/burst_${m}_$rep.json 2> $O/burst_${m}_$rep.err
  python3 -c "
import json
d = json.loads([l for l in open('$O/burst_${m}_$rep.json') if l.startswith('{')][-1]); print('burst', '$m', 'run', $rep, d['value'], d['ms_per_step'], flush=True)"
done
done
