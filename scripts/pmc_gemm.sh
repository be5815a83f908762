#!/bin/bash
# pmc_gemm.sh VARIANT : LDS / issue counters of one 4096^3 GEMM under rocprofv3 (separate passes per counter group)
v=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_v${v}_$i -o p --output-format csv -- python3 $R/scripts/gemm_one.py 0 4096 4096 4096 $v 3 > /dev/null 2>&1 || { echo "pass $i failed"; exit 1; }
  f=$(find $R/gpurun_out/pmc_v${v}_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm256' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    print(f'{k:36s} {sum(v)/len(v):16.0f}   (n={len(v)})')
PY
done
