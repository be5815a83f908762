"""rocprofv3 *_kernel_stats.csv -> a markdown table (top kernels, ms per step).  usage: stats_md.py CSV STEPS 'title' 'command' > OUT.md"""
import csv, sys
path, steps, title, cmd = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# {title}\n')
print(f'Command (on the GPU box): `{cmd}`')
print(f'({steps} profiled train steps; total kernel time {tot / 1e6:.1f} ms = {tot / 1e6 / steps:.1f} ms/step of summed kernel durations.)\n')
print('| kernel | calls | avg us | ms/step | % |\n|---|---:|---:|---:|---:|')
for r in rows[:28]:
    name = r['Name'] if len(r['Name']) < 90 else r['Name'][:87] + '...'
    print(f"| `{name}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.2f} | {float(r['Percentage']):.2f} |")
