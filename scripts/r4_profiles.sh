#!/bin/bash
# r4_profiles.sh : round-4 rocprofv3 evidence on one box: kernel stats of the default (bf16) bench command (side streams on / single stream) and of
# the fp32 one; PMC passes over the fp32 attention kernels (attention_x3.hip) through scripts/attn_one.py
R=${GRAFT_REPO_ROOT:-.}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o $tag -- python3 $R/bench.py "$@" > $O/prof_$tag.log 2>&1; tail -1 $O/prof_$tag.log | cut -c1-160; find $O/prof_$tag -name "*kernel_stats.csv" | head -1; }
prof r4 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --roofline-steps 0
LECCR_DW_STREAM=0 LECCR_TEXT_STREAM=0 LECCR_CAP_STREAM=0 prof r4_ss --steps 3 --warmup 1 --no-cpu-baseline --no-parity --roofline-steps 0
prof r4_fp32 --dtype fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --roofline-steps 0
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  DT=f32 rocprofv3 --pmc $grp -d $O/pmc_x3attn_$i -o p --output-format csv -- python3 $R/scripts/attn_one.py 2 > /dev/null 2>&1 || { echo "pass $i failed ($grp)"; continue; }
  f=$(find $O/pmc_x3attn_$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY' | tee -a $O/r4_pmc_x3attn.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    for key in ('fwd_kernel', 'dkv_kernel', 'dq_kernel'):
        if key in n and 'x3a' in n:
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in acc.items():
    for k, v in d.items():
        print(f'{key:12s} {k:34s} {sum(v)/len(v):16.0f}   (n={len(v)})')
PY
done
