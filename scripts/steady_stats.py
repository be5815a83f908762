"""rocprofv3 *_kernel_trace.csv -> per-step kernel table of the STEADY-STATE steps only.
    steady_stats.py TRACE.csv 'title' 'command' > OUT.md
The stats CSV of a bench run divides setup work (parameter arena copies, first-step allocations, fp8 site priming) by the step count:
round 2's "188 copyBuffer per step" were 750 setup copies / 4.  Here the window runs from the start of the second step to the start of the
last one (a step starts with its patchify launch), and everything is divided by the number of steps inside it."""
import collections, csv, sys
path, title, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[4] if len(sys.argv) > 4 else 'patchify_kernel'      # a kernel launched exactly once per step (the image patch gather)
marks = [int(r['Start_Timestamp']) for r in rows if marker in r['Kernel_Name']]
assert len(marks) >= 3, 'need at least 3 steps in the trace'
lo, hi, steps = marks[1] - 1, marks[-1] - 1, len(marks) - 2              # from the start of step 2 to the start of the last step
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    s = int(r['Start_Timestamp'])
    if lo <= s < hi:
        a = agg[r['Kernel_Name']]
        a[0] += 1
        a[1] += (int(r['End_Timestamp']) - s) / 1e6
tot = sum(v[1] for v in agg.values())
print(f'# {title}\n')
print(f'Command (on the GPU box): `{cmd}`')
print(f'({steps} steady-state train steps = {(hi - lo) / 1e6 / steps:.1f} ms wall per step under the profiler; summed kernel durations '
      f'{tot / steps:.1f} ms/step; setup and the first step excluded.)\n')
print('| kernel | launches/step | avg us | ms/step | % |\n|---|---:|---:|---:|---:|')
for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:36]:
    nm = name if len(name) < 90 else name[:87] + '...'
    print(f'| `{nm}` | {n / steps:.1f} | {ms / n * 1e3:.1f} | {ms / steps:.2f} | {100 * ms / tot:.2f} |')
torch_ms = sum(v[1] for k, v in agg.items() if 'at::native' in k or 'rocclr' in k)
torch_n = sum(v[0] for k, v in agg.items() if 'at::native' in k or 'rocclr' in k)
print(f'\ntorch / runtime kernels (at::native::*, __amd_rocclr_*): {torch_n / steps:.0f} launches, {torch_ms / steps:.3f} ms per step')
for name, (n, ms) in sorted(((k, v) for k, v in agg.items() if 'at::native' in k or 'rocclr' in k), key=lambda kv: -kv[1][1])[:8]:
    print(f'  {n / steps:6.1f} x {ms / n * 1e3:7.1f} us = {ms / steps:.3f} ms/step  {name[:110]}')
