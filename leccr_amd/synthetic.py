"""Synthetic Multi30K-shaped batches: what the reference's DataLoader + tokenizer hand to
RetrievalModel.forward (image_Retrieval_caption.py:42-63), generated from a seed instead of JPEGs and captions."""
import torch

CLS, SEP = 101, 102


class Encoding:
    """The two attributes of a HF BatchEncoding that the model reads."""

    def __init__(self, input_ids, attention_mask):
        self.input_ids, self.attention_mask = input_ids, attention_mask

    def to(self, device):
        return Encoding(self.input_ids.to(device), self.attention_mask.to(device))


def _tokens(g, b, L, vocab):
    ids = torch.randint(min(1000, vocab // 2), vocab, (b, L), generator=g)
    cls, sep = (CLS, SEP) if vocab > SEP else (1, 2)       # tiny test vocabularies
    lens = torch.randint(min(8, L), L + 1, (b,), generator=g)
    lens[0] = L                                     # padding='longest': at least one full-length row
    pos = torch.arange(L)[None, :]
    mask = (pos < lens[:, None]).long()
    ids[:, 0] = cls
    ids[torch.arange(b), lens - 1] = sep
    return Encoding(ids * mask, mask)


def make_batch(b, rank=0, res=384, L_t=32, L_c=64, vocab=119547, seed=42):
    """image ~ N(0,1) f32 [b,3,res,res]; src/tgt ids (L_t) and generated-caption ids (L_c) with [CLS] .. [SEP] pad;
    idx = arange + rank*b with 1 in 16 rows duplicated (soft-label branch, xvlm.py:283-290)."""
    g = torch.Generator().manual_seed(seed + rank)
    image = torch.randn(b, 3, res, res, generator=g)
    src, tgt, cap = _tokens(g, b, L_t, vocab), _tokens(g, b, L_t, vocab), _tokens(g, b, L_c, vocab)
    idx = torch.arange(b) + rank * b
    for i in range(15, b, 16):
        idx[i] = idx[i - 1]
    return dict(image=image, text=[src, tgt], captions=cap, idx=idx)


def make_video_batch(b, rank=0, frames=8, width=512, L_t=32, L_c=64, vocab=119547, seed=42):
    """SURVEY 8(d) video inputs: frame features ~ N(0,1) [b,frames,width]; 1 in 8 clips ragged (3..frames real frames,
    bool mask True = real); texts / captions / idx as make_batch."""
    g = torch.Generator().manual_seed(seed + rank)
    video = torch.randn(b, frames, width, generator=g)
    lens = torch.full((b,), frames)
    for i in range(7, b, 8):
        lens[i] = int(torch.randint(3, frames + 1, (1,), generator=g))
    mask = torch.arange(frames)[None, :] < lens[:, None]
    src, tgt, cap = _tokens(g, b, L_t, vocab), _tokens(g, b, L_t, vocab), _tokens(g, b, L_c, vocab)
    idx = torch.arange(b) + rank * b
    for i in range(15, b, 16):
        idx[i] = idx[i - 1]
    return dict(video=video, video_mask=mask, text=[src, tgt], captions=cap, idx=idx)


MULTI30K_CFG = {'embed_dim': 256, 'temp': 0.07, 'use_one_cl_proj_only': False, 'caption_encoder_name': 'mbert',
                'num_queries': 4, 'caption_ca_layer': 3, 'caption_interaction_layer': 2, 'weight_reg_loss': 0.01,
                'weight_caption_loss': 0.01, 'weight_dstl_loss': 0.5, 'weight_cv_loss': 0.01,
                'optimizer': {'opt': 'adamW', 'lr': 1e-5, 'weight_decay': 0.01, 'lr_mult': 2},
                'schedular': {'sched': 'linear', 'lr': 1e-5, 'epochs': 50, 'num_warmup_steps': 0.1}}

# VATEX-shaped video configuration (BASELINE configs[3]; frame features of width 512, 8 frames)
VATEX_CFG = dict(MULTI30K_CFG, vision_width=512, vision_layer=4)
