#!/bin/bash
# fp8 steps with the round-4 fp8 GEMM changes (k-step schedules by measurement, bias sums inside the fp8 weight-gradient launches):
# fp8 tests, then config E / config B fp8 lines: previous build, this build without the fused bias sums, this build
set -e
R=$PWD; O=$R/gpurun_out/fp8step; mkdir -p $O
python -m pytest tests -q -x -m gpu -k "fp8" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
line() { tag=$1; shift; "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag FAILED"; tail -5 $O/$tag.err; }; python3 - <<PY
import json
try:
    d = json.loads([l for l in open('$O/$tag.json') if l.startswith('{')][-1]); p = d.get('parity') or {}
    print('$tag', d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'logits', p.get('max_abs_logits_err'), 'ranks', p.get('gt_ranks_equal_to_oracle_encoders'), flush=True)
except Exception as e:
    print('$tag', 'no json', e, flush=True)
PY
}
E="--clip ViT-L/14 --text xlmr-large --batch 256 --dtype fp8 --steps 6 --warmup 2 --no-cpu-baseline --no-parity --roofline-steps 0"
B="--dtype fp8 --steps 8 --warmup 3 --no-cpu-baseline --no-parity --roofline-steps 0"
line E_old env LECCR_LIB=$R/gpurun_variants/libleccr_presink.so LECCR_FP8_BIAS_ASUM=0 python3 bench.py $E
line E_new_noasum env LECCR_FP8_BIAS_ASUM=0 python3 bench.py $E
line E_new python3 bench.py $E
line E_old2 env LECCR_LIB=$R/gpurun_variants/libleccr_presink.so LECCR_FP8_BIAS_ASUM=0 python3 bench.py $E
line E_new2 python3 bench.py $E
line B_old env LECCR_LIB=$R/gpurun_variants/libleccr_presink.so LECCR_FP8_BIAS_ASUM=0 python3 bench.py $B
line B_new python3 bench.py $B
