#!/bin/bash
# which kernels the vendor library runs on the step's NT shapes (names carry the macro tile), with their launch geometry and registers
set -e
R=$PWD
mkdir -p gpurun_out/vendor
cd /tmp && export TMPDIR=/tmp
LIBCMP=1 REPS=5 ONLY="qkv,out,fc1,fc2,sq 8192" rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vendor -o vt -- python3 $R/scripts/gemm_bench.py > $R/gpurun_out/vendor/bench.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/vendor/**/vt_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'][:200], r['Grid_Size_X'], r['Grid_Size_Y'], r['Workgroup_Size_X'], r.get('LDS_Block_Size',''), r.get('VGPR_Count',''), r.get('Accum_VGPR_Count',''), r.get('SGPR_Count',''), r.get('Scratch_Size',''))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(k, []).append(d)
with open('gpurun_out/vendor/summary.txt', 'w') as o:
    for k, v in agg.items():
        v2 = sorted(v)
        o.write(f'{len(v):4d} x med {v2[len(v2)//2]:9.1f} us  grid {k[1]}x{k[2]} wg {k[3]} lds {k[4]} vgpr {k[5]} agpr {k[6]} sgpr {k[7]} scratch {k[8]}  {k[0]}\n')
print(open('gpurun_out/vendor/summary.txt').read())
PY
cat gpurun_out/vendor/bench.log
