#!/bin/bash
# r5_profiles2.sh : the round's closing rocprofv3 evidence for the final build on one box: the vendor yardstick, kernel stats of the default bench command
# (side streams on / single stream), the GEMM family's fabric traffic (PMC)
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd $R
{ echo "== default dispatch (8-wave NT with non-temporal stores, column groups for wide outputs, 4-wave TN) + vendor library"; LIBCMP=1 REPS=10 python3 scripts/gemm_bench.py; } 2>&1 | grep -v "amdgpu.ids" > $O/r5_yardstick2.txt
tail -14 $O/r5_yardstick2.txt
bash scripts/profile_bench.sh r5 && bash scripts/profile_bench.sh r5_ss LECCR_DW_STREAM=0 LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0 && ROUND=05 bash scripts/pmc_traffic.sh
