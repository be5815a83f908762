#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_video -o v -- python3 $R/scripts/video_real_geometry.py bf16 > $R/gpurun_out/prof_video.log 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$R/gpurun_out/prof_video/v_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms (3 steps)', tot / 1e6)
for r in rows[:18]:
    print(f"{r['Name'][:95]:95s} {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us tot {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
PY
