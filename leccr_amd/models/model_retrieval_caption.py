"""RetrievalModel (image) with the reference's API (models/model_retrieval_caption.py:9-199) over the HIP ops."""
import torch
import torch.nn as nn

from .. import diag, ops
from .attention import CrossAttention
from .layers import Linear
from .xvlm import XVLMBase, _dist_info, local_rows


class RetrievalModel(XVLMBase):
    def __init__(self, config):
        super().__init__(config, load_vision_params=False, load_text_params=False, use_contrastive_loss=True,
                         use_matching_loss=True, use_mlm_loss=False, use_bbox_loss=False)
        self.init_params = []          # as the reference (model_retrieval_caption.py:14): nothing gets lr * lr_mult
        self.caption_encoder_name = config['caption_encoder_name']
        self.weight_caption_loss = config['weight_caption_loss']
        self.n_queries = config['num_queries']
        self.caption_ca_layer = config['caption_ca_layer']
        self.caption_interaction_layer = config['caption_interaction_layer']
        self.weight_reg_loss = config['weight_reg_loss']
        self.weight_cv_loss = config['weight_cv_loss']
        self.weight_dstl_loss = config['weight_dstl_loss']

    def init_caption_encoder(self):
        """model_retrieval_caption.py:33-53.  The caption encoder IS the text encoder (aliased module)."""
        if self.caption_encoder_name == 'mbert':
            self.caption_encoder = self.text_encoder
            self.caption_width = self.text_width
        elif self.caption_encoder_name == 'clip':      # CLIP's text tower encodes the captions (forward only, never trained)
            self.caption_width = self.vision_width
        else:
            print('caption_width error')
            exit()
        vw = self.vision_width
        self.caption_proj = Linear(self.caption_width, vw)
        self.queries = nn.Parameter(torch.zeros(self.n_queries, 1, vw))
        self.crossattn_query = CrossAttention(d_model=vw, nhead=8, num_layers=self.caption_ca_layer)
        self.crossattn = CrossAttention(d_model=vw, nhead=8, num_layers=self.caption_interaction_layer)
        self.crossattn2 = CrossAttention(d_model=vw, nhead=8, num_layers=self.caption_interaction_layer)
        self.caption_proj1 = Linear(vw, self.embed_dim)
        self.cproj = Linear(vw, vw)
        self.vproj = Linear(vw, vw)

    def get_caption_embeds(self, text_ids, text_atts=None):
        """model_retrieval_caption.py:55-63."""
        if self.caption_encoder_name == 'mbert':
            self._ensure_runtime()
            return self.caption_encoder(text_ids, attention_mask=text_atts, return_dict=True).last_hidden_state
        if self.caption_encoder_name == 'clip':
            self._ensure_runtime()
            return self.clip_encoder.encode_text(text_ids, return_hidden=True)[-1]
        print('caption_encoder_name error!')
        exit()

    # ------------------------------------------------------------------ batch-major internals
    def _interaction(self, image_embeds, caption_hidden, key_padding_mask):
        b = image_embeds.size(0)
        queries = self.queries.permute(1, 0, 2).expand(b, -1, -1).to(image_embeds.dtype).contiguous()
        cap = self.caption_proj(caption_hidden)
        ori_caption = self.crossattn_query(queries, cap, key_padding_mask)
        after_image = self.crossattn(image_embeds, ori_caption, None)
        after_caption = self.crossattn2(ori_caption, image_embeds, None)
        return after_image, after_caption, ori_caption

    def _cv_loss(self, ori_caption, ori_image, idx):
        """caption_vision_loss in pooled form: the mean over both token axes is linear, so
        sim[i,j] = <mean_c chat_i, mean_v vhat_j> with chat/vhat normalised over the TOKEN axis (F.normalize's default
        dim=1).  Project + normalise + pool locally, all-gather 2 x [b, vw] instead of the token tensors.  cproj/vproj sit
        after the gather in the reference, so under DDP their gradients are the FULL dL/dtheta: wgrad_scale = W."""
        return self._cv_loss_pooled(*self._cv_pooled(ori_caption, ori_image), idx)

    def _cv_pooled(self, ori_caption, ori_image):
        _, world = _dist_info()
        return (ops.CvPoolFn.apply(self.cproj(ori_caption, wgrad_scale=float(world))),
                ops.CvPoolFn.apply(self.vproj(ori_image, wgrad_scale=float(world))))

    def _cv_loss_pooled(self, pc, pv, idx):
        gather_rows = self.gathers.rows
        idx_all = gather_rows(idx.view(-1, 1)).view(-1).contiguous()
        return ops.SimInfoNCEFn.apply(gather_rows(pc), gather_rows(pv), None, idx_all, False, 1.0, local_rows(pc.size(0)))

    # ------------------------------------------------------------------ reference API (LND at the boundary)
    def caption_regularization(self, caption_embeds):
        """model_retrieval_caption.py:65-73; caption_embeds [n, b, d] (LND)."""
        return ops.CapRegFn.apply(caption_embeds.transpose(0, 1))

    def interaction_with_caption(self, image_embeds, caption_embeds, key_padding_mask):
        """model_retrieval_caption.py:76-84 -> (after_image [N,b,vw], after_caption [n,b,vw], caption [n,b,vw]) LND."""
        self._ensure_runtime()
        a_img, a_cap, o_cap = self._interaction(image_embeds, caption_embeds, key_padding_mask)
        return a_img.transpose(0, 1).contiguous(), a_cap.transpose(0, 1).contiguous(), o_cap.transpose(0, 1).contiguous()

    def dstl_loss(self, image_embeds, caption_embeds, text_embeds_s, text_embeds_t, idx, alpha=0.8):
        """model_retrieval_caption.py:94-116; caption_embeds [n, b, D] (LND)."""
        gather_rows = self.gathers.rows
        v = gather_rows(image_embeds.float())
        c = gather_rows(caption_embeds.transpose(0, 1).float().contiguous())
        s, t = gather_rows(text_embeds_s.float()), gather_rows(text_embeds_t.float())
        return ops.DstlFn.apply(v, c, s, t, alpha, local_rows(image_embeds.size(0)))

    def caption_vision_loss(self, caption, image, idx):
        """model_retrieval_caption.py:118-143; caption [n, b, vw] (LND), image [b, N, vw]."""
        return self._cv_loss(caption.transpose(0, 1).contiguous(), image, idx)

    def get_caption_contrastive_loss(self, caption_embeds, text_feats, tval=None):
        """model_retrieval_caption.py:145-152; caption_embeds [n, b, D] (LND); local batch only."""
        return ops.CapSimFn.apply(caption_embeds.transpose(0, 1).float().contiguous(), text_feats.float(), self.temp, tval)

    def forward(self, image, text_ls, captions=None, idx=None, epoch=None, cap_idx=None):
        """model_retrieval_caption.py:154-199 -> (loss_itc_vs + loss_cv, loss_itc_vt', loss_itc_st, w_c loss_itc_c,
        w_r loss_reg_c).  Same five scalars; internally batch-major, src+tgt text in one BERT pass."""
        self._ensure_runtime()
        self.gathers.begin()
        gather_rows = self.gathers.rows
        ops.begin_step()
        if self.fp8_sites is not None:         # scales of this step = what the previous step's tensors asked for (delayed scaling)
            self.fp8_sites.update()
        if self.training:          # fresh dropout masks per step and per rank (the reference seeds with seed + rank)
            self._step_count = getattr(self, '_step_count', 0) + 1
            rank, _ = _dist_info()
            ops.set_dropout_seed((int(self.config.get('seed', 42)) + rank) * 1000003 + self._step_count)
        tval = None           # the loss kernels read self.temp on the device (round 2: float(self.temp) here drained the stream)
        ids_s, atts_s = text_ls[0].input_ids, text_ls[0].attention_mask
        ids_t, atts_t = text_ls[1].input_ids, text_ls[1].attention_mask
        b = ids_s.shape[0]
        L = max(ids_s.shape[1], ids_t.shape[1])

        def pad(x):
            return x if x.shape[1] == L else torch.nn.functional.pad(x, (0, L - x.shape[1]))
        if self.caption_encoder_name == 'clip':     # captions = CLIP token ids [b, context_length], padding id 0 (model_retrieval_caption.py:158-167)
            cap_ids = captions
            cap_atts = (captions != 0).to(torch.int64)
        else:
            cap_ids, cap_atts = captions.input_ids, captions.attention_mask

        def caption_side():
            with torch.no_grad():
                return self.get_caption_embeds(cap_ids, cap_atts)

        def text_side():
            hid = self.get_text_embeds(torch.cat([pad(ids_s), pad(ids_t)], 0), torch.cat([pad(atts_s), pad(atts_t)], 0))
            return self.get_features(None, hid)             # padded key positions are masked: [CLS] rows are unchanged

        # The image tower first (program order = the order the reference runs them; the text towers' autograd nodes are the
        # younger ones, so backward enqueues them first), the two BERT passes beside it on the text stream when enabled.
        ori_image, _ = self.get_vision_embeds(image)
        ts = ops.text_stream(image.device)
        if ts is None:
            cap_hidden, feats = caption_side(), text_side()
        else:
            main = torch.cuda.current_stream()
            cs = ops.caption_stream(image.device) or ts      # the forward-only caption pass on a stream of its own
            cs.wait_stream(main)
            ts.wait_stream(main)
            with torch.cuda.stream(cs):
                cap_hidden = caption_side()
            with torch.cuda.stream(ts):
                feats = text_side()
            main.wait_stream(cs)
            main.wait_stream(ts)
            cap_hidden.record_stream(main)
            feats.record_stream(main)
        kpm = ~cap_atts.bool()
        after_image, after_caption, ori_caption = self._interaction(ori_image, cap_hidden, kpm)

        pc, pv = self._cv_pooled(ori_caption, ori_image)
        loss_reg_c = ops.CapRegFn.apply(ori_caption)

        image_feat = self.get_features(after_image, None)
        text_feat_s, text_feat_t = feats[:b], feats[b:]
        cq = self.caption_proj1(after_caption, out_f32=True).contiguous()   # [b, n, D] f32
        # every block the loss terms gather (W > 1), in one collective instead of sixteen
        diag.begin('loss_stage_fwd')
        diag.begin('allgather')
        self.gathers.packed([image_feat, text_feat_s, text_feat_t, cq, pc, pv], None if idx is None else idx.view(-1, 1))
        diag.end('allgather')
        loss_cv = self._cv_loss_pooled(pc, pv, idx) * self.weight_cv_loss

        loss_itc_vs = self.get_contrastive_loss(image_feat, text_feat_s, idx=idx, tval=tval)
        loss_itc_vt = self.get_contrastive_loss(image_feat, text_feat_t, idx=idx, tval=tval)
        loss_itc_st = self.get_contrastive_loss(text_feat_s, text_feat_t, idx=idx, tval=tval)

        loss_itc_c = ops.CapSimFn.apply(cq, text_feat_s, self.temp, tval) + \
            ops.CapSimFn.apply(cq, text_feat_t, self.temp, tval)

        loss_dstl = ops.DstlFn.apply(gather_rows(image_feat), gather_rows(cq), gather_rows(text_feat_s),
                                     gather_rows(text_feat_t), 0.8, local_rows(image_feat.size(0))) * self.weight_dstl_loss
        loss_itc_vt = loss_itc_vt * (1 - self.weight_dstl_loss) + loss_dstl
        diag.end('loss_stage_fwd')
        return (loss_itc_vs + loss_cv, loss_itc_vt, loss_itc_st, loss_itc_c * self.weight_caption_loss,
                loss_reg_c * self.weight_reg_loss)
