#!/bin/bash
# r5_profiles.sh : round-5 rocprofv3 evidence on one box: the vendor yardstick (8-wave kernel with non-temporal stores, 4-wave n4 kernel), kernel stats of
# the default bench command (side streams on / single stream), the GEMM family's fabric traffic (PMC), matrix-pipe counters of both NT kernels on qkv / fc1
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
{ echo "== default dispatch (8-wave NT with non-temporal stores, 4-wave TN) + vendor library"; LIBCMP=1 REPS=10 python3 scripts/gemm_bench.py; echo "== NT products on the 4-wave n4 kernel (variant 5)"; VARIANT=5 REPS=10 ONLY="qkv,out,fc1,fc2,bert qkv,bert fc1,bert fc2,sq 4096,sq 8192" python3 scripts/gemm_bench.py; } 2>&1 | grep -v "amdgpu.ids" > $O/r5_yardstick.txt
tail -30 $O/r5_yardstick.txt
bash scripts/profile_bench.sh r5
bash scripts/profile_bench.sh r5_ss LECCR_DW_STREAM=0 LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0
ROUND=05 bash scripts/pmc_traffic.sh
SHAPES=nt2 VARIANT=2 TAG=w8 bash scripts/pmc_gemm_shapes.sh > /dev/null
SHAPES=nt2 VARIANT=5 TAG=n4 bash scripts/pmc_gemm_shapes.sh > /dev/null
cat $O/pmc_gemm_shapes_w8.txt $O/pmc_gemm_shapes_n4.txt | head -80
