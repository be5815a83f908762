#!/bin/bash
# pmc_gemm_shapes.sh : issue / MFMA / LDS counters of the 256x256 GEMM on the step's own shapes (round-1 VERDICT item 3: "commit --pmc
# passes for the step's shapes"), one shape and one counter group per rocprofv3 pass (no trace domains beside --pmc).
# Writes gpurun_out/pmc_gemm_shapes.txt.  SQ_* counters are summed over the chip; "per launch" = average over the profiled launches
# after the first.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=147712
OUT=$R/gpurun_out/pmc_gemm_shapes${TAG:+_$TAG}.txt
: > $OUT
while read name tn m n k; do
  echo "== $name  tn=$tn M=$m N=$n K=$k" >> $OUT
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    i=$((i+1))
    d=$R/gpurun_out/pmc_gs_${name}_$i
    rm -rf $d
    rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 $R/scripts/gemm_one.py $tn $m $n $k ${VARIANT:-0} 4 > /dev/null 2>&1 || { echo "pass $i failed" >> $OUT; continue; }
    f=$(find $d -name "*counter_collection.csv" | head -1)
    python3 - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm256' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    w = v[1:] if len(v) > 1 else v
    print(f'  {k:34s} {sum(w)/len(w):18.0f}   (launches {len(w)})')
PY
  done
done <<LIST
$( if [ "$SHAPES" = "nt2" ]; then printf 'qkv 0 %s 2304 768\nfc1 0 %s 3072 768\n' $M $M; elif [ "$SHAPES" = "dw" ]; then printf 'dW_qkv 1 2304 768 %s\ndW_fc1 1 3072 768 %s\ndW_fc2 1 768 3072 %s\n' $M $M $M; else printf 'qkv 0 %s 2304 768\nfc1 0 %s 3072 768\nfc2 0 %s 768 3072\nout 0 %s 768 768\ndW_fc1 1 3072 768 %s\n' $M $M $M $M $M; fi )
LIST
cat $OUT
