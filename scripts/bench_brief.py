"""Print the headline fields of a bench.py JSON line: python scripts/bench_brief.py gpurun_out/bench.log"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d['roofline']
print(f"{d['value']} {d['unit']}  {d['ms_per_step']} ms/step  roofline {r['frac']} ({r['avg_launch_us']} us/launch)  gemm {r.get('gemm_ms_per_step')} ms  "
      f"attn fwd {r.get('attn_fwd_ms_per_step')} bwd {r.get('attn_bwd_ms_per_step')} ms  parity {d.get('parity', {}).get('ok')} "
      f"{d.get('parity', {}).get('max_abs_loss_err')}  cpu {d.get('cpu_baseline', {}).get('value')}")
for k in d.get('hbm_kernels', []):
    print(f"   {k['kernel']:16s} {k['ms_per_step']:7.3f} ms  {k['achieved_tb_s']:.2f} TB/s")
