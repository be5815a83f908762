"""The video model at the reference's MSR-VTT geometry (configs/cclm-base-ft/Retrieval_msrvtt.yaml:36-37: 4096-wide frame features, one
SelfAttention layer): three synthetic train steps of batch 32 on one MI355X.  usage: video_real_geometry.py [bf16|fp32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from leccr_amd.models.video_model_retrieval_caption import RetrievalModel
from leccr_amd.synthetic import VATEX_CFG, make_video_batch
from leccr_amd.optim import create_optimizer
cfg = dict(VATEX_CFG, vision_width=4096, vision_layer=1, compute_dtype=sys.argv[1] if len(sys.argv) > 1 else 'bf16')
torch.manual_seed(0)
model = RetrievalModel(cfg); model.init_caption_encoder(); model = model.to('cuda').train()
opt = create_optimizer(dict(lr=1e-4, weight_decay=0.01), model)
b = make_video_batch(32, frames=8, width=4096)
dev = 'cuda'
to = lambda x: x.to(dev)
import time
for step in range(3):
    t0 = time.time()
    losses = model(to(b['video']), to(b['video_mask']), [t.to(dev) for t in b['text']], captions=b['captions'].to(dev), idx=to(b['idx']))
    opt.zero_grad(); sum(losses).backward(); opt.step(); torch.cuda.synchronize()
    print('step', step, [round(float(l), 4) for l in losses], round(time.time() - t0, 3), 's', flush=True)
print('params', sum(p.numel() for p in model.parameters()) / 1e6, 'M')
