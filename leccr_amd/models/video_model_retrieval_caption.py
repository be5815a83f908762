"""RetrievalModel (video) with the reference's API (models/video_model_retrieval_caption.py:9-227) over the HIP ops.

Differences from the image model, all taken from the reference: the vision tower is a SelfAttention stack over
pre-extracted frame features `[b, F, vision_width]` with a frame mask (no CLIP encoder, XVLMBase_video); crossattn2 masks
the padded frames; the visual feature is the MASKED MEAN of the interacted frame tokens (xvlm_video.py:260-272) instead of
the CLS token; caption_vision_loss stays on the local batch (its all-gathers are commented out upstream).
"""
import torch

from .. import ops
from .attention import SelfAttention
from .model_retrieval_caption import RetrievalModel as _ImageRetrievalModel
from .xvlm import _dist_info, local_rows


class RetrievalModel(_ImageRetrievalModel):
    _with_clip_encoder = False

    def __init__(self, config):
        super().__init__(config)
        self.vision_width = config['vision_width']
        self.vid_encoder = SelfAttention(d_model=self.vision_width, nhead=8, num_layers=config['vision_layer'])

    # ------------------------------------------------------------------ reference API
    def get_vision_embeds(self, video, mask):
        """video_model_retrieval_caption.py:30-34: (frame tokens [b,F,vw], key padding mask = ~mask)."""
        self._ensure_runtime()
        kpm = ~mask.bool()
        return self.vid_encoder(video.to(self.compute_dtype).contiguous(), kpm), kpm

    def get_features(self, image_embeds=None, text_embeds=None, vis_pooling='mean', vis_mask=None):
        """xvlm_video.py:260-277: text = CLS -> Linear -> normalise; video = masked mean over frames -> Linear -> normalise.
        vis_mask [b, F, 1] (True = real frame), as the reference passes it."""
        vision_proj = self.text_proj if self.vision_proj is None else self.vision_proj
        if image_embeds is None:
            return self._feat(self.text_proj, text_embeds[:, 0, :])
        assert text_embeds is None and vis_pooling == 'mean' and vis_mask is not None
        m = vis_mask.to(torch.float32)
        pooled = (image_embeds.float() * m).sum(1) / m.sum(1)                # tiny [b,F,vw] elementwise: plain torch ops
        return self._feat(vision_proj, pooled.to(image_embeds.dtype))

    def _interaction(self, image_embeds, caption_hidden, key_padding_mask, video_kpm=None):
        b = image_embeds.size(0)
        queries = self.queries.permute(1, 0, 2).expand(b, -1, -1).to(image_embeds.dtype).contiguous()
        cap = self.caption_proj(caption_hidden)
        ori_caption = self.crossattn_query(queries, cap, key_padding_mask)
        after_video = self.crossattn(image_embeds, ori_caption, None)
        after_caption = self.crossattn2(ori_caption, image_embeds, video_kpm)        # padded frames masked as keys
        return after_video, after_caption, ori_caption

    def interaction_with_caption(self, image_embeds, caption_embeds, key_padding_mask, video_mask):
        """video_model_retrieval_caption.py:96-108 -> three LND tensors; video_mask is the KEY PADDING mask (True = pad)
        that get_vision_embeds returned, as in the reference's forward."""
        self._ensure_runtime()
        a_vid, a_cap, o_cap = self._interaction(image_embeds, caption_embeds, key_padding_mask, video_mask)
        return a_vid.transpose(0, 1).contiguous(), a_cap.transpose(0, 1).contiguous(), o_cap.transpose(0, 1).contiguous()

    def _cv_loss(self, ori_caption, ori_image, idx):
        """caption_vision_loss on the LOCAL batch (video_model_retrieval_caption.py:144-169), pooled form."""
        pc = ops.CvPoolFn.apply(self.cproj(ori_caption))
        pv = ops.CvPoolFn.apply(self.vproj(ori_image))
        return ops.SimInfoNCEFn.apply(pc, pv, None, idx.view(-1).contiguous(), False, 1.0)

    def forward(self, video, video_mask, text_ls, captions=None, idx=None, epoch=None, cap_idx=None):
        """video_model_retrieval_caption.py:180-227 -> the same five scalars as the image model."""
        self._ensure_runtime()
        self.gathers.begin()
        gather_rows = self.gathers.rows
        ops.begin_step()
        if self.fp8_sites is not None:         # scales of this step = what the previous step's tensors asked for (delayed scaling)
            self.fp8_sites.update()
        if self.training:
            self._step_count = getattr(self, '_step_count', 0) + 1
            rank, _ = _dist_info()
            ops.set_dropout_seed((int(self.config.get('seed', 42)) + rank) * 1000003 + self._step_count)
        tval = None           # the loss kernels read self.temp on the device
        ori_video, video_kpm = self.get_vision_embeds(video, video_mask)
        with torch.no_grad():
            cap_ids, cap_atts = captions.input_ids, captions.attention_mask
            cap_hidden = self.get_caption_embeds(cap_ids, cap_atts)
        after_video, after_caption, ori_caption = self._interaction(ori_video, cap_hidden, ~cap_atts.bool(), video_kpm)

        loss_cv = self._cv_loss(ori_caption, ori_video, idx) * self.weight_cv_loss
        loss_reg_c = ops.CapRegFn.apply(ori_caption)

        image_feat = self.get_features(after_video, None, vis_mask=video_mask.unsqueeze(-1))
        ids_s, atts_s = text_ls[0].input_ids, text_ls[0].attention_mask
        ids_t, atts_t = text_ls[1].input_ids, text_ls[1].attention_mask
        b = ids_s.shape[0]
        L = max(ids_s.shape[1], ids_t.shape[1])

        def pad(x):
            return x if x.shape[1] == L else torch.nn.functional.pad(x, (0, L - x.shape[1]))
        hid = self.get_text_embeds(torch.cat([pad(ids_s), pad(ids_t)], 0), torch.cat([pad(atts_s), pad(atts_t)], 0))
        feats = self.get_features(None, hid)
        text_feat_s, text_feat_t = feats[:b], feats[b:]

        loss_itc_vs = self.get_contrastive_loss(image_feat, text_feat_s, idx=idx, tval=tval)
        loss_itc_vt = self.get_contrastive_loss(image_feat, text_feat_t, idx=idx, tval=tval)
        loss_itc_st = self.get_contrastive_loss(text_feat_s, text_feat_t, idx=idx, tval=tval)

        cq = self.caption_proj1(after_caption, out_f32=True)
        loss_itc_c = ops.CapSimFn.apply(cq, text_feat_s, self.temp, tval) + \
            ops.CapSimFn.apply(cq, text_feat_t, self.temp, tval)
        loss_dstl = ops.DstlFn.apply(gather_rows(image_feat), gather_rows(cq.contiguous()), gather_rows(text_feat_s),
                                     gather_rows(text_feat_t), 0.8, local_rows(image_feat.size(0))) * self.weight_dstl_loss
        loss_itc_vt = loss_itc_vt * (1 - self.weight_dstl_loss) + loss_dstl
        return (loss_itc_vs + loss_cv, loss_itc_vt, loss_itc_st, loss_itc_c * self.weight_caption_loss,
                loss_reg_c * self.weight_reg_loss)
