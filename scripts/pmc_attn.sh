#!/bin/bash
# pmc_attn.sh : issue / LDS / MFMA counters of the ViT attention kernels under rocprofv3 (separate passes per counter group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_attn_$i -o p --output-format csv -- python3 $R/scripts/attn_one.py 2 > /dev/null 2>&1 || { echo "pass $i failed ($grp)"; continue; }
  f=$(find $R/gpurun_out/pmc_attn_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    for key in ('flash_fwd', 'flash_dkdv', 'flash_dq'):
        if key in n:
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in acc.items():
    for k, v in d.items():
        print(f'{key:12s} {k:34s} {sum(v)/len(v):16.0f}   (n={len(v)})')
PY
done
