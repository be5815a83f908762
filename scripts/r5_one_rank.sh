#!/bin/bash
# round 5: the RCCL calls of the N > 1 path issued on the one-GPU box (a one-rank 'nccl' world, LECCR_ONE_RANK_WORLD=1):
# the tests, then the default workload (batch 256) inside such a world beside the plain line - what the collective path costs before any link is involved
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_rccl_one_rank_gpu.py -q -m gpu -x > gpurun_out/r5_one_rank.log 2>&1
rc=$?
tail -5 gpurun_out/r5_one_rank.log
[ $rc -ne 0 ] && exit $rc
A="--steps 10 --warmup 3 --no-cpu-baseline --no-reference-precision --no-parity --roofline-steps 0"
for cus in 0 16; do
  RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29850 + cus)) LECCR_ONE_RANK_WORLD=1 LECCR_RCCL_CUS=$cus \
    timeout -k 10 300 python bench.py $A > gpurun_out/r5_one_rank_b256_cus$cus.json 2> gpurun_out/r5_one_rank_b256_cus$cus.err || exit 1
done
LECCR_BENCH_DIAG=1 timeout -k 10 300 python bench.py $A > gpurun_out/r5_one_rank_b256_plain.json 2> gpurun_out/r5_one_rank_b256_plain.err || exit 1
python - <<'PY'
import json
for t in ('cus0', 'cus16', 'plain'):
    o = json.loads(open(f'gpurun_out/r5_one_rank_b256_{t}.json').read().strip().splitlines()[-1])
    print(t, o['value'], o['ms_per_step'], o['config']['backend'], o['config']['rccl_version'], o.get('stage_ms_per_rank'))
PY
