#!/bin/bash
# round 4: new parity tests, eval bench mode, 2-rank gloo rehearsal with the stage timers
python -m pytest tests/test_model_gpu.py -q -k "full_frame_geometry" 2>&1 | tail -4 && \
python bench.py --mode eval --steps 2 --warmup 1 > gpurun_out/r4_eval_a.json 2> gpurun_out/r4_eval_a.err; tail -2 gpurun_out/r4_eval_a.err; cat gpurun_out/r4_eval_a.json
python bench.py --mode eval --steps 2 --warmup 1 --eval-batch 64 --eval-text-batch 256 --eval-texts 5000 > gpurun_out/r4_eval_b.json 2>> gpurun_out/r4_eval_a.err; cat gpurun_out/r4_eval_b.json
LECCR_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 2 --warmup 1 --batch 32 --no-cpu-baseline > gpurun_out/r4_gloo2.json 2> gpurun_out/r4_gloo2.err; tail -2 gpurun_out/r4_gloo2.err
python -c "
import json;d=json.load(open('gpurun_out/r4_gloo2.json'));print(d['value'],d.get('stage_ms_per_rank'))"
