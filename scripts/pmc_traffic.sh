#!/bin/bash
# pmc_traffic.sh : HBM bytes per launch of the 256x256 GEMM family inside bench.py, from separate rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads; KB units)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
DT=${DT:-bf16}                       # DT=fp32: the fp32 line's GEMM family (-> r${ROUND:-05}_gemm256_hbm_traffic_fp32.json)
SUF=$([ "$DT" = bf16 ] && echo "" || echo "_$DT")
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/pmc_traffic${SUF}_$c -o t --output-format csv -- python3 $R/bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-reference-precision --roofline-steps 0 > /dev/null 2>&1 || { echo "pass $c failed"; exit 1; }
done
python3 - <<PY
import csv, glob, json
def total(c):
    f = glob.glob('$R/gpurun_out/pmc_traffic${SUF}_%s/**/*counter_collection.csv' % c, recursive=True)[0]
    s, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if 'gemm256' in r['Kernel_Name'] and r['Counter_Name'] == c:        # gemm256_kernel (8-wave NT / TN), gemm256t4_kernel (4-wave TN)
            s += float(r['Counter_Value']); n += 1
    return s, n
fs, nf = total('FETCH_SIZE'); ws, nw = total('WRITE_SIZE')
assert nf == nw and nf > 0, (nf, nw)
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --dtype $DT --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-reference-precision --roofline-steps 0 (scripts/pmc_traffic.sh)",
       "kernel_family": "gemm256 family (gemm256_kernel: 8-wave NT / TN instantiations; gemm256t4_kernel: 4-wave TN weight gradient)", "launches": nf, "fetch_size_kb_sum": fs, "write_size_kb_sum": ws,
       "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
       "hbm_bytes_per_launch": (2 * fs + ws) * 1024 / nf}
json.dump(out, open('$R/gpurun_out/r${ROUND:-05}_gemm256_hbm_traffic${SUF}.json', 'w'), indent=1)
print(json.dumps(out)[:400])
PY
