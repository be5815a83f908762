"""BASELINE configs[3] as worded - "8 frames x ViT-B/16, double-sim loss" - composed end to end.

The reference's video model starts at PRE-EXTRACTED per-frame feature vectors (models/video_model_retrieval_caption.py:26-34,
dataset/__init__.py:143-162); the extractor that produced them is not in its repository.  This class is that front end
put in front of the same model, trained jointly: every frame goes through the CLIP ViT (clip/model.py:223-240,340-346), its
[CLS] feature (ln_post + proj, width vw) is the frame feature, and the frame sequence feeds the reference video RetrievalModel
unchanged (`vid_encoder` SelfAttention over frames, masked-mean feature, double-sim losses).  An extension beyond the
reference, built only from reference-defined pieces; its parity test composes the oracle's `encode_image` with its
`video_retrieval_forward` (tests/test_model_gpu.py::test_frame_vit_video_model_vs_oracle).
"""
import torch

from .. import ops
from .video_model_retrieval_caption import RetrievalModel as _VideoRetrievalModel
from .xvlm import build_clip_encoder


class FrameViTRetrievalModel(_VideoRetrievalModel):
    def __init__(self, config):
        clip, vw = build_clip_encoder(dict(config, clip_text_tower=False))
        super().__init__(dict(config, vision_width=vw))
        self.clip_encoder = clip          # 'clip_encoder.visual.*' = the arena's late region: its backward runs last (ops._fire)

    def encode_frames(self, frames):
        """frames [b, F, 3, R, R] f32 -> frame features [b, F, vw] (the [CLS] row of ln_post(x) @ proj per frame)."""
        self._ensure_runtime()
        b, F = frames.shape[:2]
        v = self.clip_encoder.visual
        x = v(frames.reshape(b * F, *frames.shape[2:]).contiguous(), self.compute_dtype)
        cls = ops.LnProjFn.apply(x[:, 0:1, :].contiguous(), v.ln_post.weight, v.ln_post.bias, v.proj)
        return cls.view(b, F, -1)

    def forward(self, frames, video_mask, text_ls, captions=None, idx=None, epoch=None, cap_idx=None):
        return super().forward(self.encode_frames(frames), video_mask, text_ls, captions=captions, idx=idx, epoch=epoch,
                               cap_idx=cap_idx)
