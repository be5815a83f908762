#!/bin/bash
# kernel trace of the default workload inside a one-rank RCCL world: what the collective path launches beside the plain step
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_one_rank_prof
mkdir -p $OUT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29871 LECCR_ONE_RANK_WORLD=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o one -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-reference-precision --roofline-steps 0 > $OUT/one.json 2> $OUT/one.err
rc=$?
cd $GRAFT_REPO_ROOT
python scripts/steady_stats.py $OUT/one_kernel_trace.csv "one-rank RCCL world, config B bf16" "scripts/r5_one_rank_prof.sh" > $OUT/one_steady.md 2>&1
ls $OUT | head; grep -i -E "nccl|rccl|memcpy|copy" $OUT/one_kernel_stats.csv | head -20
exit $rc
