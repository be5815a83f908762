"""Evaluation side of the path: evaluation_coarse and itm_eval with the reference's signatures
(image_Retrieval_caption.py:83-163, 261-317), the score matrix and the ranking computed on the GPU.

itm_eval replaces the host loop of per-row `np.argsort(score)[::-1]` + `np.where(inds == gt)` with one launch of
leccr_rank_of_gt (rank = number of strictly larger scores in the row), which is the same integer for tie-free rows, and
returns the same 13-key dict, bit for bit.
"""
import time
import datetime

import numpy as np
import torch
import torch.distributed as dist

from . import kernels as K
from .parallel import collective_path


def _ranks(scores, gt_lists, device):
    s = torch.as_tensor(np.asarray(scores) if not torch.is_tensor(scores) else scores, dtype=torch.float32)
    s = s.to(device).contiguous()
    per = max(len(g) for g in gt_lists)
    gt = torch.full((s.shape[0], per), -1, dtype=torch.int32)
    for i, g in enumerate(gt_lists):
        gt[i, :len(g)] = torch.as_tensor(list(g), dtype=torch.int32)
    return K.rank_of_gt(s, gt.to(device)).cpu().numpy().astype(np.float64)


@torch.no_grad()
def itm_eval(scores_i2t, scores_t2i, txt2img, img2txt, device='cuda'):
    """image_Retrieval_caption.py:261-317.  img2txt: {image index: [text indices]}, txt2img: {text index: image index}."""
    n_img, n_txt = (scores_i2t.shape[0], scores_t2i.shape[0])
    ranks = _ranks(scores_i2t, [img2txt[i] for i in range(n_img)], device)
    tr1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    tr5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    tr10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    ranks = _ranks(scores_t2i, [[txt2img[t]] for t in range(n_txt)], device)
    ir1 = 100.0 * len(np.where(ranks < 1)[0]) / len(ranks)
    ir5 = 100.0 * len(np.where(ranks < 5)[0]) / len(ranks)
    ir10 = 100.0 * len(np.where(ranks < 10)[0]) / len(ranks)
    tr_mean = (tr1 + tr5 + tr10) / 3
    ir_mean = (ir1 + ir5 + ir10) / 3
    r_mean = (tr_mean + ir_mean) / 2
    txt_sumr = tr1 + tr5 + tr10
    img_sumr = ir1 + ir5 + ir10
    sumr_avg = np.round((txt_sumr + img_sumr) / 6, 2)
    return {'txt_r1': tr1, 'txt_r5': tr5, 'txt_r10': tr10, 'txt_r_mean': tr_mean, 'txt_sum_r': txt_sumr,
            'img_r1': ir1, 'img_r5': ir5, 'img_r10': ir10, 'img_r_mean': ir_mean, 'r_mean': r_mean,
            'img_sumr': img_sumr, 'sumr_avg': sumr_avg, 'sumr_sum': (txt_sumr + img_sumr)}


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _gather_rows_varlen(rows, index, total):
    """Every rank holds some rows of a [total, D] matrix (`rows` [k, D] with their global row numbers `index` [k]);
    returns the complete matrix on every rank.  One size exchange + two padded all-gathers (RCCL on 'nccl')."""
    rank, world = _world()
    D = rows.shape[1]
    k = torch.tensor([rows.shape[0]], device=rows.device, dtype=torch.int64)
    sizes = [torch.zeros_like(k) for _ in range(world)]
    dist.all_gather(sizes, k)
    kmax = int(max(int(x) for x in sizes))
    pad_r = torch.zeros(kmax, D, device=rows.device, dtype=rows.dtype)
    pad_i = torch.full((kmax,), -1, device=rows.device, dtype=torch.int64)
    pad_r[:rows.shape[0]] = rows
    pad_i[:rows.shape[0]] = index
    all_r = [torch.empty_like(pad_r) for _ in range(world)]
    all_i = [torch.empty_like(pad_i) for _ in range(world)]
    dist.all_gather(all_r, pad_r)
    dist.all_gather(all_i, pad_i)
    out = torch.empty(total, D, device=rows.device, dtype=rows.dtype)
    for r, i, n in zip(all_r, all_i, sizes):
        n = int(n)
        out[i[:n]] = r[:n]
    return out


class GraphedForward:
    """fn(*inputs) -> tensor for inputs of FIXED shapes, captured as a HIP graph (torch.cuda.CUDAGraph: every launch of this library goes to
    torch's current stream and is captured as it is) and replayed per batch.  What it is for: evaluation as the reference batches it
    (image_Retrieval_caption.py:100-141: batch_size_test = 6 images, batch_size_test_text = 32 texts per forward) is ~150 launches of 10 - 60 us per
    ViT layer pair on 3462 token rows - bound by the HOST's launch rate (958 images/s), and half of the chip idles under each launch (14 row tiles).
    A replay costs the host one call, and `instances` copies of the graph with their own buffers replay on their own streams, so consecutive
    batches overlap on the chip.  The result of a call is valid once join() has run (it makes the caller's stream wait for every instance)."""

    def __init__(self, fn, example, instances=None):
        import os
        instances = instances or int(os.environ.get('LECCR_EVAL_GRAPH_INSTANCES', '2'))
        self.main = torch.cuda.current_stream()
        self.inst, self.calls = [], 0
        for _ in range(instances):
            st = torch.cuda.Stream()
            ins = [t.clone() for t in example]
            st.wait_stream(self.main)
            with torch.cuda.stream(st):
                fn(*ins)                                  # once outside the capture: lazy initialisation (function attributes, allocator pools)
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st, capture_error_mode='thread_local'):      # (a loader's pin-memory thread may allocate meanwhile)
                out = fn(*ins)
            self.inst.append((st, g, ins, out))

    def __call__(self, *inputs):
        st, g, ins, out = self.inst[self.calls % len(self.inst)]
        self.calls += 1
        cur = torch.cuda.current_stream()
        st.wait_stream(cur)                               # the inputs were produced on the caller's stream
        with torch.cuda.stream(st):
            for a, b in zip(ins, inputs):
                assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
                a.copy_(b, non_blocking=True)
                b.record_stream(st)                       # (the caller may drop b at once: its block must outlive this copy)
            g.replay()
            res = out.clone()                             # the instance's output buffer is rewritten by its next replay
        res.record_stream(cur)
        return res

    def join(self):
        cur = torch.cuda.current_stream()
        for st, _, _, _ in self.inst:
            cur.wait_stream(st)


def eval_graphs_on(config, device):
    """As-yielded evaluation batches run as replayed HIP graphs (config['eval_graphs'], default on; LECCR_EVAL_GRAPHS=0 / the in-library event
    profiler's pass in bench.py - its records come from the launches themselves - turn it off)."""
    import os
    return bool(config.get('eval_graphs', True)) and os.environ.get('LECCR_EVAL_GRAPHS', '1') != '0' and torch.device(device).type == 'cuda'


def _graph_for(model, kind, fn, example):
    """The captured forward of `kind` ('text' / 'image') for inputs shaped like `example`, kept on the model between evaluation passes (the kernels
    read the weights' compute copies through fixed pointers: an optimizer step in between rewrites them in place)."""
    cache = model.__dict__.setdefault('_eval_graphs', {})
    key = (kind, next(model.parameters()).data_ptr()) + tuple((tuple(t.shape), t.dtype) for t in example)
    if key not in cache:
        try:
            cache[key] = GraphedForward(fn, example)
        except RuntimeError as e:          # a forward that cannot be captured (a host read inside it, ...): say so once and run launch by launch
            print('evaluation: HIP graph capture of the %s forward failed (%s); running it launch by launch' % (kind, str(e).splitlines()[0][:200]))
            torch.cuda.synchronize()
            cache[key] = None
    return cache[key]


def eval_group_sizes(config, n_images, num_text, world):
    """Rows per forward of the two evaluation loops (config['eval_group_images'] / ['eval_group_texts'], default 256 / 1024; 0 = as the loader
    yields them).  The groups are also the unit of the rank sharding (group c -> rank c % world), so under W > 1 they are capped at a 1 / W
    share of the set: the 1000 images / 1000 - 5000 texts of the Multi30K / MSCOCO-1k test sets then give EVERY rank of an 8-GPU node a group
    (at 256 / 1024 rows there were 4 image groups and 1 - 5 text chunks - half the ranks or more encoded nothing)."""
    gi, gt = int(config.get('eval_group_images', 256)), int(config.get('eval_group_texts', 1024))
    if world > 1:
        if gi > 0 and n_images > 0:
            gi = max(1, min(gi, -(-n_images // world)))
        if gt > 0 and num_text > 0:
            gt = max(1, min(gt, -(-num_text // world)))
    return gi, gt


@torch.no_grad()
def evaluation_coarse(model, data_loader, tokenizer, device, config, shard=True):
    """image_Retrieval_caption.py:83-163: text features in chunks of batch_size_test_text, image features through the
    caption interaction, score = image_embeds @ text_embeds^T (one HIP GEMM, f32).  Returns (i2t, t2i) numpy arrays.

    Under torch.distributed the reference makes EVERY rank encode every text and image and then all-reduces the (already
    complete) score matrix, which only scales it by W.  Here (shard=True) rank r encodes text chunks and image batches
    r, r+W, r+2W, ...; the embeddings are all-gathered once and every rank forms the same score matrix: 1/W of the encoder
    work per rank, identical scores."""
    model.eval()
    rank, world = _world()
    if not shard:
        rank, world = 0, 1
    print('Computing features for evaluation...')
    start_time = time.time()
    texts = data_loader.dataset.text
    num_text = len(texts)
    # The reference encodes batch_size_test_text (32) texts and batch_size_test (6) images per forward (its YAMLs): 3462 token rows per ViT
    # GEMM, a launch-bound pass on this chip.  Nothing in eval mode depends on how the rows are batched, so several loader batches are run
    # through the towers together (config['eval_group_images'] / ['eval_group_texts'], default 256 / 1024 rows per forward; 0 = as the
    # loader yields them): same scores, same ranks, 5.8 x the images per second (958 as yielded, 4224 at 64 / 256, 5100 at 128 / 512, 5543).
    try:
        n_images = len(data_loader.dataset)
    except TypeError:
        n_images = 0
    gi, gt = eval_group_sizes(config, n_images, num_text, world)
    try:
        n_batches = len(data_loader)                                  # (a capture pays for itself after a few replays)
    except TypeError:
        n_batches = 0
    text_bs = max(config['batch_size_test_text'], gt)
    text_embeds, text_rows = [], []
    graphs = eval_graphs_on(config, device) and not getattr(model, 'fp8', False)      # (fp8: the sites' host-side state is not part of a capture)
    graphed = []                                                      # GraphedForward objects of this pass (joined before their results are read)

    def text_fwd(ids, mask):
        return model.get_features(text_embeds=model.get_text_embeds(ids, mask))

    for c, i in enumerate(range(0, num_text, text_bs)):
        if c % world != rank:
            continue
        text = texts[i: min(num_text, i + text_bs)]
        ti = tokenizer(text, padding='max_length', truncation=True, max_length=config['max_tokens'],
                       return_tensors='pt').to(device)
        if graphs and gt == 0 and len(text) == text_bs and num_text >= 4 * text_bs:      # as yielded: whole batches replay one captured forward
            text_graph = _graph_for(model, 'text', text_fwd, (ti.input_ids, ti.attention_mask))
            if text_graph is not None and text_graph not in graphed:
                graphed.append(text_graph)
            text_embeds.append((text_graph or text_fwd)(ti.input_ids, ti.attention_mask))
        else:
            text_embeds.append(text_fwd(ti.input_ids, ti.attention_mask))
        text_rows.append(torch.arange(i, i + len(text), device=device))
    image_embeds, image_rows = [], []
    image_graph = [None]

    def image_fwd(image, cap_ids, cap_mask):
        image_feat, _ = model.get_vision_embeds(image)
        caption_embed = model.get_caption_embeds(cap_ids, cap_mask)
        image_feat, _, _ = model.interaction_with_caption(image_embeds=image_feat, caption_embeds=caption_embed,
                                                          key_padding_mask=~cap_mask.bool())
        return model.get_features(image_embeds=image_feat.transpose(0, 1).contiguous())

    def encode_images(image, generated_captions, first):
        image = image.to(device)
        image_feat, _ = model.get_vision_embeds(image)
        if config.get('caption_encoder_name', 'mbert') == 'clip':       # image_Retrieval_caption.py:128-136: CLIP token ids, padding id 0
            if torch.is_tensor(generated_captions):
                cap_ids = generated_captions.to(device)
            else:
                from clip import tokenize as clip_tokenize                 # the reference's package (BPE vocabulary ships with it)
                cap_ids = clip_tokenize(generated_captions, truncate=True).to(device)
            caption_embed = model.get_caption_embeds(cap_ids, None)
            key_padding_mask = cap_ids == 0
        else:
            ci = tokenizer(generated_captions, padding='max_length', truncation=True, max_length=config['max_tokens'],
                           return_tensors='pt').to(device)
            if graphs and gi == 0 and n_batches >= 4:       # as yielded (batch_size_test images): whole batches replay one captured forward
                if image_graph[0] is None:                                   # the first batch sets the shape; a shorter last batch runs as it is
                    image_graph[0] = _graph_for(model, 'image', image_fwd, (image, ci.input_ids, ci.attention_mask)) or False
                    if image_graph[0]:
                        graphed.append(image_graph[0])
                g = image_graph[0]
                if g and all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(g.inst[0][2], (image, ci.input_ids, ci.attention_mask))):
                    image_embeds.append(g(image, ci.input_ids, ci.attention_mask))
                    image_rows.append(torch.arange(first, first + image.shape[0], device=device))
                    return
            caption_embed = model.get_caption_embeds(ci.input_ids, ci.attention_mask)
            key_padding_mask = ~ci.attention_mask.bool()
        image_feat, _, _ = model.interaction_with_caption(image_embeds=image_feat, caption_embeds=caption_embed,
                                                          key_padding_mask=key_padding_mask)
        image_embeds.append(model.get_features(image_embeds=image_feat.transpose(0, 1).contiguous()))
        image_rows.append(torch.arange(first, first + image.shape[0], device=device))

    n_img, group, pend_i, pend_c, pend_first = 0, 0, [], [], 0
    for image, generated_captions, img_id in data_loader:
        if not pend_i:
            pend_first = n_img
        n_img += image.shape[0]
        pend_i.append(image)
        pend_c.append(generated_captions)
        if sum(t.shape[0] for t in pend_i) >= max(gi, 1):
            if group % world == rank:
                caps = torch.cat(pend_c) if torch.is_tensor(pend_c[0]) else [c for cs in pend_c for c in cs]
                encode_images(torch.cat([t.to(device) for t in pend_i]) if len(pend_i) > 1 else pend_i[0], caps, pend_first)
            group, pend_i, pend_c = group + 1, [], []
    if pend_i:
        if group % world == rank:
            caps = torch.cat(pend_c) if torch.is_tensor(pend_c[0]) else [c for cs in pend_c for c in cs]
            encode_images(torch.cat([t.to(device) for t in pend_i]) if len(pend_i) > 1 else pend_i[0], caps, pend_first)

    for g in graphed:
        g.join()

    def assemble(chunks, rows, total):
        D = model.embed_dim
        e = torch.cat(chunks, dim=0).float() if chunks else torch.zeros(0, D, device=device)
        r = torch.cat(rows) if rows else torch.zeros(0, device=device, dtype=torch.int64)
        return e.contiguous() if not collective_path(world) else _gather_rows_varlen(e.contiguous(), r, total)

    text_embeds = assemble(text_embeds, text_rows, num_text)
    image_embeds = assemble(image_embeds, image_rows, n_img)
    score_matrix_i2t = K.gemm(image_embeds.contiguous(), text_embeds.contiguous())
    score_matrix_t2i = score_matrix_i2t.t().contiguous()
    print('Evaluation time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    return score_matrix_i2t.cpu().numpy(), score_matrix_t2i.cpu().numpy()


def _norm_score(s):
    """video_Retrieval_caption_double_sim.py norm_score: (s - min) / max(s - min), over the whole matrix."""
    s = s - s.min()
    return s / s.max()


@torch.no_grad()
def evaluation_coarse_video(model, data_loader, tokenizer, device, config, alpha=0.9):
    """video_Retrieval_caption_double_sim.py:94-190 ("double sim"): score = alpha * norm(video_feat . text_feat) +
    (1 - alpha) * norm(max over the n caption queries of caption_feat . text_feat).  data_loader yields
    (video [b,F,vw], video_mask [b,F] bool, generated captions, ids).  Both products are HIP GEMMs (f32); the reference's
    closing all_reduce of the complete matrices only scales them by W (ranking-invariant) and is dropped."""
    model.eval()
    print('Computing features for evaluation...')
    start_time = time.time()
    texts = data_loader.dataset.text
    text_bs = config['batch_size_test_text']
    text_embeds = []
    for i in range(0, len(texts), text_bs):
        ti = tokenizer(texts[i: i + text_bs], padding='max_length', truncation=True, max_length=config['max_tokens'],
                       return_tensors='pt').to(device)
        text_embeds.append(model.get_features(text_embeds=model.get_text_embeds(ti.input_ids, ti.attention_mask)))
    text_embeds = torch.cat(text_embeds, dim=0).float().contiguous()
    image_embeds, caption_embeds = [], []
    for video, mask_video, generated_captions, vid_id in data_loader:
        video, mask_video = video.to(device), mask_video.to(device)
        vis, kpm = model.get_vision_embeds(video, mask_video)
        ci = tokenizer(generated_captions, padding='max_length', truncation=True, max_length=config['max_tokens'],
                       return_tensors='pt').to(device)
        cap = model.get_caption_embeds(ci.input_ids, ci.attention_mask)
        a_vid, a_cap, _ = model.interaction_with_caption(image_embeds=vis, caption_embeds=cap,
                                                         key_padding_mask=~ci.attention_mask.bool(), video_mask=kpm)
        image_embeds.append(model.get_features(image_embeds=a_vid.transpose(0, 1).contiguous(),
                                               vis_mask=mask_video.unsqueeze(-1)))
        caption_embeds.append(model.caption_proj1(a_cap.transpose(0, 1).contiguous(), out_f32=True))      # [b, n, D]
    image_embeds = torch.cat(image_embeds, dim=0).float().contiguous()
    caption_embeds = torch.cat(caption_embeds, dim=0).float().contiguous()
    nv, n, D = caption_embeds.shape
    score = K.gemm(image_embeds, text_embeds)                                   # [Nv, Nt]
    c_score = K.gemm(caption_embeds.view(nv * n, D), text_embeds).view(nv, n, -1).max(dim=1)[0]
    i2t = alpha * _norm_score(score) + (1.0 - alpha) * _norm_score(c_score)
    print('Evaluation time {}'.format(str(datetime.timedelta(seconds=int(time.time() - start_time)))))
    return i2t.cpu().numpy(), i2t.t().contiguous().cpu().numpy()
