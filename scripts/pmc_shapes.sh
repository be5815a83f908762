#!/bin/bash
# pmc_shapes.sh : fabric-side bytes per launch of the 256x256 GEMM on each of the step's shapes, one shape per process, from separate
# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE x2 on gfx950 for 16 B/lane
# streaming reads), against the algorithmic bytes of the shape (every operand read once, the output written once; the accumulating
# f32 weight-gradient output read + written).  Writes gpurun_out/pmc_shapes.txt.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
M=147712
OUT=$R/gpurun_out/pmc_shapes.txt
: > $OUT
# name tn M N K
while read name tn m n k; do
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$R/gpurun_out/pmc_shape_${name}_$c
    rm -rf $d
    rocprofv3 --pmc $c --kernel-trace -d $d -o t --output-format csv -- python3 $R/scripts/gemm_one.py $tn $m $n $k 0 4 > /dev/null 2>&1 || { echo "pass $name $c failed" >> $OUT; continue; }
  done
  python3 - "$name" $tn $m $n $k "$R" >> $OUT <<'PY'
import csv, glob, sys
name, tn, m, n, k, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
def per_launch(c):
    f = glob.glob(f'{R}/gpurun_out/pmc_shape_{name}_{c}/**/*counter_collection.csv', recursive=True)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'gemm256_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
    return sum(v[1:]) / max(1, len(v) - 1), len(v)          # first launch dropped (cold caches)
fs, nf = per_launch('FETCH_SIZE'); ws, nw = per_launch('WRITE_SIZE')
rd = 2 * fs * 1024; wr = ws * 1024
alg_rd = (m * k + n * k) * 2 + (m * n * 4 if tn else 0)
alg_wr = m * n * (4 if tn else 2)
print(f'{name:10s} tn={tn} M={m:6d} N={n:5d} K={k:6d}  read {rd/1e6:8.1f} MB (alg {alg_rd/1e6:8.1f}, x{rd/alg_rd:4.2f})  write {wr/1e6:8.1f} MB (alg {alg_wr/1e6:8.1f}, x{wr/alg_wr:4.2f})  launches {nf}')
PY
done <<LIST
qkv 0 $M 2304 768
out 0 $M 768 768
fc1 0 $M 3072 768
fc2 0 $M 768 3072
dx_qkv 0 $M 768 2304
dW_qkv 1 2304 768 $M
dW_out 1 768 768 $M
dW_fc1 1 3072 768 $M
dW_fc2 1 768 3072 $M
LIST
cat $OUT
