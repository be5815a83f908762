#!/bin/bash
# which part of the collective path costs the step time?  one-rank RCCL world, phases on / off, batch 32 (host-bound) and 256
cd "$GRAFT_REPO_ROOT"
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 LECCR_ONE_RANK_WORLD=1
for b in 32 256; do
A="--batch $b --steps 20 --warmup 3 --roofline-steps 0 --no-cpu-baseline --no-reference-precision --no-parity"
for ph in 1 0 1 0; do
  MASTER_PORT=$((29890 + RANDOM % 50)) LECCR_DDP_PHASES=$ph timeout -k 10 300 python bench.py $A 2> gpurun_out/r5_orh2.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b phases $ph', o['value'], o['ms_per_step'])" || exit 1
done
done
