#!/bin/bash
# round 5: the round-2 column-group sweep again, now that the output stores are non-temporal (they no longer push the weight panels out of the L2)
cd "$GRAFT_REPO_ROOT"
S="qkv,out,fc1,fc2"
{
for g in 0 2 3 4 6 0 3; do
  echo "== LECCR_G256_COLGROUP=$g"; LECCR_G256_COLGROUP=$g REPS=10 ONLY="$S" python scripts/gemm_bench.py
done
for g in 0 3 0 3; do
  echo "== step LECCR_G256_COLGROUP=$g"; LECCR_G256_COLGROUP=$g python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-parity --no-reference-precision 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['gemm_ms_per_step'])"
done
} 2>&1 | grep -v "amdgpu.ids\|^variant" | tee gpurun_out/r5_colgroup.txt
