#!/bin/bash
# quick variant of pmc_shapes.sh: qkv / fc1 / dpre only (read + write passes), honours LECCR_G256_COLGROUP from the environment
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=147712
while read name tn m n k; do
  for c in FETCH_SIZE; do
    d=$R/gpurun_out/pmc_nt_${name}_$c
    rm -rf $d
    rocprofv3 --pmc $c --kernel-trace -d $d -o t --output-format csv -- python3 $R/scripts/gemm_one.py $tn $m $n $k 0 4 > /dev/null 2>&1 || { echo "pass $name $c failed"; continue; }
    python3 - "$name" $m $n $k "$d" <<'PY'
import csv, glob, sys
name, m, n, k, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
f = glob.glob(f'{d}/**/*counter_collection.csv', recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'gemm256_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
rd = 2 * sum(v[1:]) / max(1, len(v) - 1) * 1024
alg = (m * k + n * k) * 2
print(f'{name:8s} read {rd/1e6:8.1f} MB (alg {alg/1e6:7.1f}, x{rd/alg:4.2f})')
PY
  done
done <<LIST
qkv 0 $M 2304 768
fc1 0 $M 3072 768
LIST
