#!/bin/bash
# host time of the exchange hooks (diag.host: they run on autograd's thread) in a one-rank RCCL world, batch 32 (host-bound step) and 256
cd "$GRAFT_REPO_ROOT"
for b in 32 256; do
A="--batch $b --steps 10 --warmup 3 --roofline-steps 0 --no-cpu-baseline --no-reference-precision --no-parity"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29881 LECCR_ONE_RANK_WORLD=1 timeout -k 10 300 python bench.py $A > gpurun_out/r5_orh_one_$b.json 2> gpurun_out/r5_orh_one_$b.err || exit 1
LECCR_BENCH_DIAG=1 timeout -k 10 300 python bench.py $A > gpurun_out/r5_orh_plain_$b.json 2> gpurun_out/r5_orh_plain_$b.err || exit 1
done
python - <<'PY'
import json
for b in (32, 256):
    for t in ('plain', 'one'):
        o = json.loads(open(f'gpurun_out/r5_orh_{t}_{b}.json').read().strip().splitlines()[-1])
        print(b, t, o['value'], o['ms_per_step'], o.get('stage_ms_per_rank'))
PY
