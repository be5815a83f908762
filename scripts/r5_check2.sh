#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "column_groups" > gpurun_out/r5_check2.log 2>&1; rc=$?; tail -3 gpurun_out/r5_check2.log
[ $rc -ne 0 ] && exit $rc
for g in -1 0 -1 0; do echo "== LECCR_G256_COLGROUP=$g (0: automatic, -1: never)"; LECCR_G256_COLGROUP=$g REPS=10 ONLY="sq 4096,sq 8192,fc1,qkv" python scripts/gemm_bench.py 2>&1 | grep -v "amdgpu.ids\|^variant"; done | tee gpurun_out/r5_colgroup_auto.txt
