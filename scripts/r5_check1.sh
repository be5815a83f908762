#!/bin/bash
# round 5: the tests added this round + the default bench line with its reference-precision leg
set -e -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_eval_gpu.py -q -k "graded_scores" > gpurun_out/r5_check1_tests.log 2>&1 || true
tail -30 gpurun_out/r5_check1_tests.log; cat gpurun_out/graded_eval_*.txt
grep -h "graded eval\|structured eval" gpurun_out/r5_check1_tests.log || true
python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err || { tail -30 gpurun_out/r5_bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r5_bench_default.json').read().strip().splitlines()[-1])
print('bf16', d['value'], d['ms_per_step'], d['roofline']['frac'])
print('reference_precision', json.dumps(d.get('reference_precision'))[:1500])
PY
