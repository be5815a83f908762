"""LayerNorm passes that also write the fp8 copy (leccr_layernorm_fwd_ex / _bwd_ex) against the plain ones, ViT-B and ViT-L shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd import kernels as K
dev = 'cuda'
def t(f, reps=10):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for rows, d in ((256 * 577, 768), (256 * 730, 1024), (8192, 1024)):
    x = torch.randn(rows, d, device=dev).bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
    g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev); dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
    sites = K.Fp8Sites(dev, 8)
    sx, sg = sites.new(K.Fp8Sites.E4M3), sites.new(K.Fp8Sites.E5M2)
    sx.quantize(x); sg.quantize(dy)            # primes both
    y, mean, rstd, _ = K.layernorm_fwd(x, g, b, 1e-5)
    nb = rows * d * 2
    tf = t(lambda: K.layernorm_fwd(x, g, b, 1e-5))
    tfe = t(lambda: K.layernorm_fwd(x, g, b, 1e-5, q8_site=sx))
    tb = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres))
    tbe = t(lambda: K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres, q8_site=sg))
    print(f'[{rows}, {d}] fwd {tf*1e3:.1f} us {2*nb/tf/1e9:.2f} TB/s | fwd + fp8 copy {tfe*1e3:.1f} us {2.5*nb/tfe/1e9:.2f} TB/s | '
          f'bwd {tb*1e3:.1f} us {4*nb/tb/1e9:.2f} TB/s | bwd + fp8 copy {tbe*1e3:.1f} us {4.5*nb/tbe/1e9:.2f} TB/s', flush=True)
# the same with the site's amax reset before every launch, as the training step does (Fp8Sites.update() at the start of a step): every wave
# then finds its row maximum above the stored amax and issues the same-address atomic
print('with the amax reset before each launch (as in the step):')
for rows, d in ((256 * 730, 1024), (16384, 1024)):
    x = torch.randn(rows, d, device=dev).bfloat16(); dy = torch.randn_like(x); dres = torch.randn_like(x)
    g = torch.ones(d, device=dev); b = torch.zeros(d, device=dev); dg = torch.zeros(d, device=dev); db = torch.zeros(d, device=dev)
    sites = K.Fp8Sites(dev, 8)
    sx, sg = sites.new(K.Fp8Sites.E4M3), sites.new(K.Fp8Sites.E5M2)
    sx.quantize(x); sg.quantize(dy)
    y, mean, rstd, _ = K.layernorm_fwd(x, g, b, 1e-5)
    tu = t(lambda: sites.update())
    def f():
        sites.update(); K.layernorm_fwd(x, g, b, 1e-5, q8_site=sx)
    def fb():
        sites.update(); K.layernorm_bwd(dy, x, g, mean, rstd, dg, db, dres=dres, q8_site=sg)
    print(f'[{rows}, {d}] update alone {tu*1e3:.1f} us | update + fwd + fp8 copy {t(f)*1e3:.1f} us | update + bwd + fp8 copy {t(fb)*1e3:.1f} us', flush=True)
