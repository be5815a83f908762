cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_loss -o loss -- python3 $R/scripts/loss_bench.py > /dev/null 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/prof_loss/loss_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us max {float(r['MaxNs'])/1e3:9.1f} tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
