#!/bin/bash
# column groups on the square yardstick shapes: the row-major walk streams the whole B through each XCD's L2 once per tile row
cd "$GRAFT_REPO_ROOT"
{
for g in 0 2 4 6 8 12 16 0 8; do
  echo "== LECCR_G256_COLGROUP=$g"; LECCR_G256_COLGROUP=$g REPS=10 ONLY="sq 4096,sq 8192,fc1" python scripts/gemm_bench.py
done
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_colgroup_sq.txt
