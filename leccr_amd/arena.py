"""ParamArena - flat HBM storage for parameters, gradients, optimizer state and low-precision compute copies.

MI355X-first replacement for the reference's per-tensor parameter handling (605 separately allocated tensors, one
optimizer kernel chain per tensor, DDP buckets re-flattened every step - image_Retrieval_caption.py:381-419):

  * every nn.Parameter's .data is a view into ONE f32 buffer; .grad is a view into ONE f32 gradient buffer, so
    zero_grad is one memset, gradient all-reduce is a handful of large RCCL calls over contiguous memory, and AdamW is
    one kernel launch over the arena (leccr_adamw);
  * in bf16 mode `refresh()` makes the compute copies once per step: one cast kernel over the whole arena plus one
    transpose-cast per 2-D weight (K-contiguous [in,out] copies so the dX GEMMs run in the NT form the MFMA kernel
    wants);
  * parameters a module declares as a group (`_arena_groups()`) are laid out adjacently so they can also be addressed
    as one packed matrix (BERT's separate query/key/value weights -> one [3d,d] QKV operand) without changing the
    state_dict keys.

state_dict keys, named_parameters() and load_state_dict() are untouched (load_state_dict copies into the views).
"""
import torch

from . import kernels as K

_ALIGN = 64  # elements; keeps every tensor 256-byte aligned in the f32 arena and 128-byte aligned in the bf16 copy


class ParamArena:
    def __init__(self, model, compute_dtype=torch.float32, device=None):
        self.compute_dtype = compute_dtype
        params, seen = [], set()

        def add(p):
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)

        groups = []
        for m in model.modules():
            if hasattr(m, '_arena_groups'):
                for gname, plist in m._arena_groups():
                    start = len(params)
                    fresh = all(id(p) not in seen for p in plist)
                    if fresh:
                        for p in plist:
                            assert p.numel() % _ALIGN == 0, 'packed group members must be multiples of 64 elements'
                            add(p)
                        groups.append((m, gname, plist, start))
        late = []                                          # parameters whose gradients are final last (the ViT trunk)
        late_prefix = getattr(model, '_arena_late_prefix', None)
        for n, p in model.named_parameters():
            if getattr(p, '_leccr_unused', False):          # inert state_dict-only parameters stay outside the arena
                continue
            if late_prefix and n.startswith(late_prefix):
                late.append(p)
            else:
                add(p)
        n_early = len(params)
        for p in late:
            add(p)
        self.params = params
        self.names = {id(p): n for n, p in model.named_parameters()}
        device = device or params[0].device
        self.offsets, off = [], 0
        group_member = {id(p) for _, _, pl, _ in groups for p in pl[1:]}
        for p in params:
            if id(p) not in group_member:      # group members after the first follow immediately (no padding)
                off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
            self.offsets.append(off)
            off += p.numel()
        self.numel = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        # [0, early_numel): gradients complete before the late region's backward starts -> their all-reduce can overlap it
        self.early_numel = self.offsets[n_early] if n_early < len(params) else self.numel
        self.data = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=device, dtype=torch.float32)
        self.lp = torch.zeros(self.numel, device=device, dtype=torch.bfloat16) if compute_dtype == torch.bfloat16 else None
        self._transposed = []
        self._split3 = []               # fp32 path: weights whose bf16 hi / lo split copies are made once per step (need_split3)
        self._tt_cache = {}
        self.post_refresh = []          # callables run after every refresh (derived copies outside the arena)
        self.touched = [False] * len(params)
        for i, (p, o) in enumerate(zip(params, self.offsets)):
            view = self.data[o:o + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
            p._arena, p._aidx = self, i
            p._g = p.grad
            p._lp = self.lp[o:o + p.numel()].view(p.shape) if self.lp is not None else p.data
            p._lpt = None
        for m, gname, plist, start in groups:
            o, n = self.offsets[start], sum(p.numel() for p in plist)
            rows = sum(p.shape[0] for p in plist)
            shape = (rows,) + tuple(plist[0].shape[1:])
            pk = _Packed()
            pk.members = plist
            pk._g = self.grad[o:o + n].view(shape)
            pk._lp = (self.lp if self.lp is not None else self.data)[o:o + n].view(shape)
            pk._lpt = None
            pk.data = self.data[o:o + n].view(shape)
            setattr(m, gname, pk)
        self.state_m = self.state_v = None
        self._seg_key = None
        self._seg_range = {}
        self.step_count = 0
        # backward-pass hooks of THIS arena's model (fired by leccr_amd.ops from the ViT backward nodes; see ops._fire)
        self.on_late_backward_start = None
        self.on_early_grads_final = None
        self.on_vit_block_backward = None

    # ------------------------------------------------------------------ compute copies
    def need_transposed(self, holder):
        """Register a 2-D weight (Parameter or packed group) whose [in,out] compute copy the dX GEMM needs (bf16 mode)."""
        if self.compute_dtype != torch.bfloat16 or holder._lpt is not None:
            return
        src = holder.data
        assert src.dim() == 2
        holder._lpt = torch.empty(src.shape[1], src.shape[0], device=src.device, dtype=torch.bfloat16)
        self._transposed.append(holder)

    def need_split3(self, holder):
        """Register a 2-D weight (Parameter or packed group) of the fp32 path whose products run on bf16 hi / lo split copies (ops.X3): the
        [out, 3 in] G-form copy of the forward product and the [in, 3 out] X-form copy of W^T of the dX product are then made ONCE per
        optimizer step, all weights in one launch (refresh), instead of by a leccr_split3 pass at every use."""
        if self.compute_dtype != torch.float32 or getattr(holder, '_s3g', None) is not None:
            return
        src = holder.data
        assert src.dim() == 2 and src.dtype == torch.float32
        R, Cn = src.shape
        holder._s3g = torch.empty(R, 3 * Cn, device=src.device, dtype=torch.bfloat16)
        holder._s3t = torch.empty(Cn, 3 * R, device=src.device, dtype=torch.bfloat16)
        self._split3.append(holder)

    def _split3_table(self, lo, hi):
        key = ('s3', lo, hi, len(self._split3))
        hit = self._tt_cache.get(key)
        if hit is None:
            base, rows, tiles = self.data.data_ptr(), [], 0
            for h in self._split3:
                off = (h.data.data_ptr() - base) // 4
                if not (lo <= off < hi):
                    continue
                R, Cn = h.data.shape
                assert h.data.is_contiguous()
                rows.append([tiles, h.data.data_ptr(), h._s3g.data_ptr(), h._s3t.data_ptr(), R, Cn])
                tiles += ((R + 63) // 64) * ((Cn + 63) // 64)
            hit = (torch.tensor(rows, dtype=torch.int64, device=self.data.device) if rows else None, tiles)
            self._tt_cache[key] = hit
        return hit

    def _transpose_table(self, lo, hi):
        """Batched-transpose table of the [in,out] copies whose master weights live in arena elements [lo, hi)."""
        key = (lo, hi, len(self._transposed))
        hit = self._tt_cache.get(key)
        if hit is None:
            base, rows, tiles = self.data.data_ptr(), [], 0
            for h in self._transposed:
                off = (h.data.data_ptr() - base) // 4
                if not (lo <= off < hi):
                    continue
                R, Cn = h.data.shape
                assert h.data.is_contiguous() and h.data.dtype == torch.float32
                rows.append([tiles, h.data.data_ptr(), h._lpt.data_ptr(), R, Cn])
                tiles += ((R + 63) // 64) * ((Cn + 63) // 64)
            hit = (torch.tensor(rows, dtype=torch.int64, device=self.data.device) if rows else None, tiles)
            self._tt_cache[key] = hit
        return hit

    @torch.no_grad()
    def refresh(self, adamw_wrote_lp=False, lo=0, hi=None):
        """Rebuild the low-precision compute copies from the f32 master weights (call once per optimizer step; [lo, hi):
        only the copies of that element range - the optimizer updates the arena in two parts when it is fused into backward)."""
        if self.lp is None:
            if getattr(self, '_split3', None):          # fp32 path: the weights' split copies (need_split3)
                table, tiles = self._split3_table(lo, self.numel if hi is None else hi)
                if table is not None:
                    K.split3_weights_batched(table, tiles)
            return
        hi = self.numel if hi is None else hi
        if not adamw_wrote_lp:
            K.cast_bf16(self.data[lo:hi], self.lp[lo:hi])
        if self._transposed:
            table, tiles = self._transpose_table(lo, hi)
            if table is not None:
                K.transpose_cast_batched(table, tiles)                      # one launch for all [in,out] copies of the range
        if hi == self.numel:
            for fn in self.post_refresh:
                fn()

    # ------------------------------------------------------------------ gradients
    def zero_grad(self):
        self.grad.zero_()

    def mark_autograd_grads(self, lo=0, hi=None):
        """Parameters updated through plain autograd (tiny ones: temp, queries) count as touched when non-zero."""
        hi = self.numel if hi is None else hi
        for i, p in enumerate(self.params):
            if lo <= self.offsets[i] < hi and not self.touched[i] and p.requires_grad and p.numel() <= 1 << 16 \
                    and bool(p.grad.abs().sum() != 0):
                self.touched[i] = True

    # ------------------------------------------------------------------ optimizer (transformers.AdamW semantics)
    def _segments(self, decay_of):
        key = tuple(self.touched)
        if self._seg_key != key:
            ends, wds = [], []
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                nxt = self.offsets[i + 1] if i + 1 < len(self.params) else self.numel
                ends.append(nxt)
                # a parameter that never received a gradient is skipped by the reference optimizer (grad is None):
                # wd < 0 tells the kernel to leave it untouched
                wds.append(decay_of(self.names[id(p)], p) if (self.touched[i] and p.requires_grad) else -1.0)
            dev = self.data.device
            self._seg_end = torch.tensor(ends, dtype=torch.int64, device=dev)
            self._seg_wd = torch.tensor(wds, dtype=torch.float32, device=dev)
            self._seg_key = key
        return self._seg_end, self._seg_wd

    def _segments_range(self, decay_of, lo, hi):
        """Segment table of arena elements [lo, hi), ends relative to lo (lo / hi are parameter boundaries)."""
        seg_end, seg_wd = self._segments(decay_of)
        if lo == 0 and hi == self.numel:
            return seg_end, seg_wd
        key = (self._seg_key, lo, hi)
        if self._seg_range.get('key') != key:
            i0 = next(i for i, o in enumerate(self.offsets) if o >= lo)
            i1 = next((i for i, o in enumerate(self.offsets) if o >= hi), len(self.offsets))
            self._seg_range = {'key': key, 'end': (seg_end[i0:i1] - lo).clamp_(max=hi - lo).contiguous(),
                               'wd': seg_wd[i0:i1].contiguous()}
        return self._seg_range['end'], self._seg_range['wd']

    @torch.no_grad()
    def adamw_step(self, lr, betas, eps, decay_of, gscale=1.0, lo=0, hi=None, step=None):
        """One AdamW update of arena elements [lo, hi) (default: everything).  `step` = the bias-correction step number of this
        update; None: the next one (and the arena's step counter advances)."""
        hi = self.numel if hi is None else hi
        if self.state_m is None:
            self.state_m = torch.zeros_like(self.data)
            self.state_v = torch.zeros_like(self.data)
        if step is None:
            self.step_count += 1
            step = self.step_count
        seg_end, seg_wd = self._segments_range(decay_of, lo, hi)
        K.adamw(self.data[lo:hi], self.grad[lo:hi], self.state_m[lo:hi], self.state_v[lo:hi], seg_end, seg_wd, lr, betas[0],
                betas[1], eps, step, gscale, lp=None if self.lp is None else self.lp[lo:hi])   # also rewrites the bf16 compute copy


class _Packed:
    """Several adjacent parameters addressed as one matrix (same attributes the kernels' callers use)."""
    members = ()
