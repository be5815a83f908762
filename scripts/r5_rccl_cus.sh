#!/bin/bash
# what a collective's resident workgroups do to the persistent GEMMs beside them, without a second GPU: a one-rank RCCL world in which every
# phased all-reduce is followed by an occupier kernel (LECCR_EMULATE_RCCL=blocks:us_per_GB: `blocks` workgroups holding their CUs for the time an
# 8-GPU ring would take), with the GEMM launches on all 256 CUs (LECCR_RCCL_CUS=0) or leaving the collective its CUs
cd "$GRAFT_REPO_ROOT"
A="--steps 12 --warmup 3 --roofline-steps 0 --no-cpu-baseline --no-reference-precision --no-parity"
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 LECCR_ONE_RANK_WORLD=1
run() { MASTER_PORT=$((29890 + RANDOM % 90)) timeout -k 10 300 python bench.py $A 2> gpurun_out/r5_rccl_cus.err | python -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=o['stage_ms_per_rank']['rank0']; print('$1', o['value'], o['ms_per_step'], 'backward', s['backward_ms'], 'exposed', s['allreduce_exposed_ms'])" || exit 1; }
run "no occupier, CUS 0            "
for us in 6000 12000; do
for b in 16 32 64; do
  LECCR_EMULATE_RCCL=$b:$us LECCR_RCCL_CUS=0 run "occupier $b blocks x $us us/GB, CUS 0 "
  LECCR_EMULATE_RCCL=$b:$us LECCR_RCCL_CUS=$b run "occupier $b blocks x $us us/GB, CUS $b"
done
done
