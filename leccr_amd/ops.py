"""Autograd layer over the HIP kernels: one torch.autograd.Function per transformer layer / stage / loss.

Design (MI355X-first, not a translation of the reference's op-by-op eager graph):
  * a Function covers a whole layer, with a hand-written backward that fuses what eager autograd cannot: residual
    gradients ride in LayerNorm-backward / GEMM epilogues, activation derivatives ride in the dX GEMM epilogue
    (LECCR_EPI_D*), packed QKV projections are addressed in place by the attention kernels;
  * parameter gradients are accumulated by the kernels straight into the ParamArena's flat gradient buffer
    (backward returns None for parameters), so there is no per-parameter AccumulateGrad traffic;
  * activations are batch-major [b, L, d] in the model's compute dtype (f32 or bf16); statistics and losses are f32.

Every tensor op here is a launch from libleccr_hip.so (kernels.py); torch supplies allocation, views and the autograd
graph only.
"""
import collections
import os

import torch

from . import kernels as K
from ._lib import EPI_DGELU, EPI_DQUICKGELU, EPI_GELU, EPI_GELU_D, EPI_MULAUX, EPI_NONE, EPI_QUICKGELU, EPI_QUICKGELU_D

_DACT = {EPI_QUICKGELU: EPI_DQUICKGELU, EPI_GELU: EPI_DGELU}
# What an MLP's first Linear saves for the backward pass beside its activation: the activation's DERIVATIVE (default, round 5) or the
# pre-activation (LECCR_ACT_DERIV=0).  The forward epilogue has sigmoid(1.702 x) / erf in registers - the derivative is four more plain
# operations there - and the dX product's epilogue becomes one multiply instead of an exp2 and a reciprocal per element behind its K loop
# (c_proj's dX at config B: 870 -> ~690 us per launch; the fp8 product of config E: 1418 -> ~1000 us).  Same bytes, nothing else reads them.
act_deriv = os.environ.get('LECCR_ACT_DERIV', '1') != '0'
_FWD_D = {EPI_QUICKGELU: EPI_QUICKGELU_D, EPI_GELU: EPI_GELU_D}


def _epi_fwd(kw):
    if act_deriv and kw.get('aux_out') is not None and kw.get('epilogue') in _FWD_D:
        kw['epilogue'] = _FWD_D[kw['epilogue']]


def _epi_bwd(kw):
    if act_deriv and kw.get('epilogue') in (EPI_DQUICKGELU, EPI_DGELU):
        kw['epilogue'] = EPI_MULAUX


# --------------------------------------------------------------------------------------- dropout plumbing
class _Rng:
    base, n = 0x5EED, 0


def set_dropout_seed(seed):
    """Start a new stream of dropout masks (call once per training step; identical on no rank: add the rank)."""
    _Rng.base, _Rng.n = int(seed) & 0x7FFFFFFFFFFFFFFF, 0


def _next_seed():
    _Rng.n += 1
    return (_Rng.base * 1000003 + _Rng.n * 7919) & 0xFFFFFFFFFFFFFFFF


# Backward-pass hooks live ON THE ARENA of the parameters whose backward is running (so a second model in the same process
# never triggers another model's optimizer / all-reduce):
#   arena.on_late_backward_start()   when the ViT backward begins (= every other parameter gradient of the step is final); set
#                                    by leccr_amd.parallel.attach_grad_sync
#   arena.on_early_grads_final()     right after it (same point of the pass); set by ArenaAdamW.enable_backward_fusion
#   arena.on_vit_block_backward(blk) at the start of each ViT block's backward; set by leccr_amd.parallel
def _fire(param, hook, *args):
    cb = getattr(getattr(param, '_arena', None), hook, None)
    if cb is not None:
        cb(*args)


def _p(module, attr):
    """Effective dropout probability of a layer: its configured p in train mode, 0 in eval mode."""
    return float(getattr(module, attr, 0.0)) if module.training else 0.0


# --------------------------------------------------------------------------------------- raw linear helpers
def _bias(b):
    return None if b is None else b.data.reshape(-1)


def _site(w, name):
    """The quantisation site `name` of a weight holder, or None outside fp8 mode."""
    return getattr(w, name, None)


# fp8 mode: LECCR_FP8_FUSE_LN=1 lets the LayerNorm passes write the fp8 copies of their outputs (leccr_layernorm_fwd_ex / _bwd_ex) instead
# of separate quantise passes over h1 / h2 / x1 and the dense layers' incoming gradients.  Measured on the config-E step (ViT-L/14 +
# XLM-R-large, batch 256): 39 ms per step less kernel time (quantise passes 294 -> 158 ms over 3 profiled steps, LayerNorm +15), wall
# 515 -> 518 ms per step while the weight-gradient products were bf16 (round 2: the backward was paced by that stream); with the fp8 dW
# product of round 3 the same switch takes the step from 474.5 to 461.3 ms and is on by default.  The steady-state fp8 test runs both settings.
fuse_fp8_copies = os.environ.get('LECCR_FP8_FUSE_LN', '1') != '0'


def _q8_site(on, rows, w, which):
    """The site under which a producer (LayerNorm pass) should write the fp8 copy of a tensor that Linear `w` consumes as its
    activation (which = '_xsite', product against w._q8) or incoming gradient ('_gsite', against w._q8t) - None when that product
    does not run on the fp8 kernel (same shape rules as _fp8_ok; the producers' outputs are contiguous)."""
    w8 = getattr(w, '_q8' if which == '_xsite' else '_q8t', None) if (on and fuse_fp8_copies) else None
    if w8 is None or not (rows >= 256 and w8.shape[0] >= 256 and w8.shape[0] % 32 == 0 and w8.shape[1] % 128 == 0):
        return None
    return getattr(w, which, None)


def _ln_fwd(x, gamma, beta, eps, site=None, x3_for=None, x3_only=False, q8_only=False, **kw):
    """K.layernorm_fwd -> (y, mean, rstd, sum or None, y8 or None): with a site (fp8 mode) the pass also writes y's fp8 copy for the
    product that consumes it - no quantise pass over y.  x3_for (mode X3: the Linear that consumes y): the pass writes y's X-form split
    copy itself (5th value) when that Linear's products run on split copies; x3_only: y then stays unwritten (None) - a pre-LN block's
    h1 / h2 are read by nothing else.  q8_only: the same for the fp8 copy (see below)."""
    if x3_for is not None and fuse_x3_copies and ln_x3_copies and K.NS == 3 and x.dtype == torch.float32 and x.is_contiguous() \
            and _x3_lin_ok(x.numel() // x.shape[-1], x3_for):
        return K.layernorm_fwd(x, gamma, beta, eps, s3_form=K.FORM_X, write_y=not x3_only, **kw)
    if site is None:
        return K.layernorm_fwd(x, gamma, beta, eps, **kw) + (None,)
    # q8_only (fp8 mode, fp8_only_ok for the consuming Linear): forward and weight-gradient product both read the e4m3 copy and nothing
    # reads the bf16 y of a pre-LN block - the pass writes the copy alone (2 of its 5 bytes per element)
    return K.layernorm_fwd(x, gamma, beta, eps, q8_site=site, write_y=not (q8_only and ln_q8_only and site.primed and x.shape[-1] % 8 == 0), **kw)


def _ln_bwd(dy, x_in, gamma, mean, rstd, dgamma, dbeta, site=None, drop=None, x3_for=None, **kw):
    """K.layernorm_bwd -> (dx, g, g8): g = dropout(dx) with drop = (p, seed), else dx - the gradient the dense layer in front of the
    residual add sees; g8 = its fp8 copy under `site` (fp8 mode, once the site has a scale), else None.
    x3_for (mode X3: the Linear whose incoming gradient g is): the pass writes g's G-form split copy as g8 - the operand of that Linear's dX and
    weight-gradient products, no leccr_split3 pass over g (24 ViT + 24 BERT launches of the fp32 step); the Linear's bias gradient then rides in the
    weight-gradient launch (asum_x3), which is why this needs _x3_asum_ok."""
    if (site is None and x3_for is not None and ln_x3_bwd_copies and dy.dtype == torch.float32 and K.f32_mfma and K.NS == 3
            and _x3_lin_ok(dy.numel() // dy.shape[-1], x3_for) and _x3_asum_ok(dy.numel() // dy.shape[-1], x3_for)):
        dx, dxd, g3 = K.layernorm_bwd(dy, x_in, gamma, mean, rstd, dgamma, dbeta, drop=drop, s3_form=K.FORM_G, **kw)
        return dx, (dxd if dxd is not None else dx), g3
    if site is None:
        r = K.layernorm_bwd(dy, x_in, gamma, mean, rstd, dgamma, dbeta, drop=drop, **kw)
        return (r[0], r[1], None) if drop is not None else (r, r, None)
    dx, dxd, g8 = K.layernorm_bwd(dy, x_in, gamma, mean, rstd, dgamma, dbeta, drop=drop, q8_site=site, **kw)
    return dx, (dxd if dxd is not None else dx), g8


# fp8 mode, ViT backward: a block's LayerNorm-1 backward writes dx, the gradient the PREVIOUS block's c_proj dX product consumes; its
# fp8 copy rides along in this one-slot hand-off, keyed by the tensor's address and shape (anything else falls back to a quantise pass)
_g8_slot = None


def _g8_give(t, t8):
    global _g8_slot
    _g8_slot = (t.data_ptr(), tuple(t.shape), t8) if t8 is not None else None


def _g8_take(t):
    global _g8_slot
    slot, _g8_slot = _g8_slot, None
    if slot is not None and slot[0] == t.data_ptr() and slot[1] == tuple(t.shape):
        return slot[2]
    return None


def _fp8_ok(x2, w8):
    """The fp8 product exists for the large Linears only (leccr_gemm: M, N >= 256, whole 128-element K-tiles, N % 32 == 0)."""
    return (x2.shape[0] >= 256 and w8.shape[0] >= 256 and w8.shape[0] % 32 == 0 and w8.shape[1] % 128 == 0 and x2.stride(1) == 1
            and x2.stride(0) % 16 == 0)


# fp8 mode: the weight-gradient products take the SAME fp8 copies as the forward / dX products of their Linear (e4m3 X saved by the
# forward, e5m2 dY of the dX product) on leccr_gemm's LECCR_FP8_BF8A trans_a = trans_b = 1 form.  LECCR_FP8_DW=0: bf16 dW as in round 2.
fp8_dw = os.environ.get('LECCR_FP8_DW', '1') != '0'
fp8_dw_launches = 0          # fp8 weight-gradient products launched (tests read it)
# Bias gradients summed inside those launches from the e5m2 copy of dY (round 4, then the default; worth 2 % of the fp8 step, and the switch the
# copy-only producers below depend on).  OFF by default since round 5: the A/B the round-4 advisor asked for (tests/test_full_geometry_gpu.py::
# test_fp8_bias_gradients_from_the_e5m2_copy_against_the_unquantised_sums, profiles/r05_fp8_bias_asum_ab.txt) shows out_proj / c_fc / c_proj bias
# gradients moving by 0.5 - 0.7 of their norm and their error against the fp32 oracle growing (0.43 -> 0.82 at worst): these column sums cancel
# to ~1/10 of a random walk's size, and e5m2 rounding (2 mantissa bits) does not.  LECCR_FP8_BIAS_ASUM=1 turns it back on.
fp8_bias_asum = os.environ.get('LECCR_FP8_BIAS_ASUM', '0') == '1'
attn_q8_only = os.environ.get('LECCR_FP8_ATTN_COPY_ONLY', '1') != '0'      # the ViT attention backward writes only the e5m2 copy of dqkv where nothing reads the bf16 one
ln_q8_only = os.environ.get('LECCR_FP8_LN_COPY_ONLY', '1') != '0'          # the ViT blocks' LayerNorm passes write only the e4m3 copy of h1 / h2 where nothing reads the bf16 one


# fp32 mode (compute_dtype 'fp32' with kernels.f32_mfma): the encoder blocks carry mode X3 in their `fp8` attribute and the SAME
# hand-over points that pass fp8 copies around pass the bf16 hi / lo split copies of leccr_split3 instead - an activation is split
# ONCE (X form: forward product and weight-gradient product), a gradient once (G form: dX product and weight-gradient product, its
# column sums = the bias gradient riding in the same pass).  Linears outside the blocks go through kernels.gemm's automatic
# splitting (each product splits its own operands).
X3 = 'x3'
fuse_x3_copies = os.environ.get('LECCR_X3_FUSE', '1') != '0'      # the GEMM epilogues write the split copies of the 4d-wide MLP tensors
ln_x3_copies = os.environ.get('LECCR_X3_LN', '1') != '0'       # .. and the encoder blocks' LayerNorm passes write h1 / h2 as split copies only
attn_x3_copies = os.environ.get('LECCR_X3_ATTN', '1') != '0'   # .. and the attention kernels write o / dqkv as the copies out_proj / in_proj read
share_x3_copies = os.environ.get('LECCR_X3_SHARE', '1') != '0'   # .. and the Linears outside the blocks make one copy per operand for all their products
# .. and a bias gradient whose dY exists only as the G-form copy a GEMM epilogue wrote (c_fc / intermediate.dense: two colsum passes over the copy's hi and
# lo segments, 48 launches and 10 ms of the fp32 step) is summed INSIDE the weight-gradient launch that reads the copy anyway (leccr_gemm_params.asum_x3);
# 2: also the bias gradients that rode in the split passes over an f32 dY (leccr_split3's colsum) - those passes become plain splits; 0: as before
x3_asum = int(os.environ.get('LECCR_X3_ASUM', '1'))
ln_x3_bwd_copies = os.environ.get('LECCR_X3_LN_BWD', '1') != '0'   # .. and the LayerNorm BACKWARD passes write the G-form copy of the gradient the next dX / dW products read


def _x3_asum_ok(g_rows, w):
    """The weight-gradient product of Linear `w` from 3-segment copies of g_rows rows runs on the 4-wave TN kernel (whose column sums can skip the
    repeated hi segment)."""
    out_f, in_f = w._lp.shape
    return x3_asum and fuse_bias_grad and K.NS == 3 and K.tn4_takes(out_f, in_f, 3 * g_rows)


def _x3_lin_ok(rows, w):
    """A Linear whose three products (forward, dX, dW) all run on the split copies: in / out features whole 64-element K-tiles."""
    out_f, in_f = w._lp.shape
    return (K.f32_mfma and w._lp.dtype == torch.float32 and w._lp.stride(1) == 1 and out_f % 64 == 0 and in_f % 64 == 0
            and rows * out_f * in_f >= K._SPLIT_MIN_WORK)


def _is_x3(t):
    return t is not None and t.dtype == torch.bfloat16


def _w_split(w, transposed):
    """The split copy of weight `w` a product of the fp32 path multiplies: [out, 3 in] in G form (forward), or [in, 3 out] = the X form of
    W^T (dX).  From the per-step copies the arena keeps (ParamArena.need_split3: one launch for all weights after each optimizer step) where
    they exist and the 3-segment form is on; else split here and now."""
    c = getattr(w, '_s3t' if transposed else '_s3g', None)
    if c is not None and K.NS == 3 and K.f32_mfma:
        return c
    return K.split3(w._lp, K.FORM_X, transpose=True) if transposed else K.split3(w._lp, K.FORM_G)


def x8_of(fp8, x2, w, x8=None):
    """The e4m3 copy of x2 that Linear `w` multiplies in fp8 mode (x8 if the producer wrote it), None when that product is bf16.
    Mode X3: the X-form split copy of the f32 x2."""
    if fp8 == X3:
        if _is_x3(x8):              # written by the producer (LayerNorm pass / GEMM epilogue): x2 itself may not exist
            return x8
        if x2 is None or x2.dtype != torch.float32 or not _x3_lin_ok(x2.shape[0], w) or x2.stride(1) != 1:
            return None
        return K.split3(x2, K.FORM_X)
    if not (fp8 and getattr(w, '_q8', None) is not None and _fp8_ok(x2, w._q8)):
        return None
    return x8 if x8 is not None else w._xsite.quantize(x2)


def g8_of(fp8, dy2, w, g8=None, bias=None):
    """The e5m2 copy of the incoming gradient dy2 of Linear `w` in fp8 mode, None when its dX product is bf16.
    Mode X3: the G-form split copy of the f32 dy2; with `bias` (the Linear's bias holder) its gradient colsum(dy2) is summed in the
    same pass and the copy is tagged (`_bias_done`) so that lin_dw does not sum it again."""
    if fp8 == X3:
        if dy2 is None or dy2.dtype != torch.float32 or not _x3_lin_ok(dy2.shape[0], w) or dy2.stride(1) != 1:
            return None
        if g8 is not None:
            return g8
        fuse = bias is not None and fuse_bias_grad and K.split3_colsum_ok(dy2) and not (x3_asum >= 2 and _x3_asum_ok(dy2.shape[0], w))
        g3 = K.split3(dy2, K.FORM_G, colsum=bias._g.reshape(-1) if fuse else None)
        if fuse:
            g3._bias_done = True
            _mark(bias)
        return g3
    if not (fp8 and getattr(w, '_q8t', None) is not None and _fp8_ok(dy2, w._q8t)):
        return None
    return g8 if g8 is not None else w._gsite.quantize(dy2)


def dw_keep(x2, x8, w):
    """What the backward pass must keep of a Linear's input x2 for its weight-gradient product: x2 itself, or - when that product will
    run from the fp8 copy x8 (fp8_dw, >= 2048 rows, in / out whole 256-tiles) - only the rows past the last whole 128-row K-tile, which
    go to the bf16 kernel (an empty tensor when there are none).  The bf16 h1 / h2 / activation tensors of a block are needed for
    nothing else, so fp8 mode holds 1 byte per element of them instead of 2 (+1).
    Mode X3 (x8 = the bf16 split copy): the weight-gradient product reads only that copy - the f32 tensor is not kept."""
    if _is_x3(x8):          # (an empty tensor of its own: a zero-row VIEW of x2 would keep x2's storage alive)
        if x2 is None:
            return x8.new_empty((0, x8.shape[1] // K.NS), dtype=torch.float32)
        return x2.new_empty((0, x2.shape[1])) if (x2.dtype == torch.float32 and _x3_lin_ok(x8.shape[0], w)) else x2
    if (fp8_dw and x8 is not None and x2.shape[0] >= 2048 and x2.shape[1] % 256 == 0 and w._q8.shape[0] % 256 == 0
            and x8.stride(0) % 16 == 0 and x8.data_ptr() % 16 == 0):
        return x2[x2.shape[0] // 128 * 128:].clone()
    return x2


def fp8_only_ok(rows, w, grad):
    """True when a tensor [rows, n] that Linear `w` consumes - as its activation (grad False) or as its incoming gradient (grad True) -
    is read ONLY through its fp8 copy: the product (forward / dX) and the weight-gradient product both run on the fp8 kernels and no
    bf16 tail is left (rows a whole number of 128-row K-tiles).  The producing GEMM then writes just the fp8 copy (write_c=False):
    2 of the 5 bytes per element of the 4d-wide activation / activation-gradient epilogues (config E: those two launches were half of
    the fp8 GEMM time, HBM-bound)."""
    if not (fp8_dw and fp8_c8_only and getattr(w, '_q8', None) is not None and rows >= 2048 and rows % 128 == 0):
        return False
    out_f, in_f = w._q8.shape
    site = w._gsite if grad else w._xsite
    return out_f % 256 == 0 and in_f % 256 == 0 and site.primed


fp8_c8_only = os.environ.get('LECCR_FP8_C8_ONLY', '1') != '0'


def lin_fwd(x2, w, b=None, fp8=False, x8=None, c8_site=None, want8=False, c8_only=False, c3_consumer=None, **kw):
    """y = x2 @ W^T (+ b) with W in nn.Linear layout [out, in]; x2 [M, in] (row stride free).
    fp8 (compute_dtype 'fp8', the encoder blocks' Linears): x2 is quantised to e4m3 under the Linear's activation site and
    multiplied with the weight's e4m3 copy on the scaled-MFMA kernel; the saved-for-backward tensors stay bf16.
    x8: x2's fp8 copy when the producer already wrote it (c8_site).  c8_site: the activation site of the Linear that CONSUMES this
    output: once that site has a scale, the epilogue also writes the output's fp8 copy under it (no separate quantise pass over
    it).  want8: return (y, y8 or None) instead of y.  c8_only (fp8_only_ok): y itself is not written, (None, y8) comes back."""
    _epi_fwd(kw)
    if fp8 == X3 and _is_x3(x8):              # f32 operands as bf16 hi / lo splits: x8 = X-form copy of x2, the weight split in G form
        w3 = _w_split(w, False)
        K.f32_split_launches += 1             # (counted like the products kernels.gemm splits by itself)
        if want8 and c3_consumer is not None and fuse_x3_copies and K.NS == 3 and _x3_lin_ok(x8.shape[0], c3_consumer):
            # the Linear that consumes this output reads only its X-form split copy (forward and weight-gradient product): the GEMM's
            # epilogue writes that copy from its registers, and where the 256 x 256 kernel takes the shape the f32 output is not
            # written at all (the 4d-wide activation: 6 instead of 4 + 4 + 6 bytes per element through HBM)
            only = K.gemm256_takes(x8.shape[0], w3.shape[0], w3.shape[1])
            return K.gemm(x8, w3, bias=_bias(b), out_dtype=torch.float32, c3_form=K.FORM_X, write_c=not only, **kw)
        y = K.gemm(x8, w3, bias=_bias(b), out_dtype=torch.float32, **kw)
        return (y, None) if want8 else y
    if fp8 and fp8 != X3 and getattr(w, '_q8', None) is not None and _fp8_ok(x2, w._q8):
        if x8 is None:
            x8 = w._xsite.quantize(x2)
        c8 = None
        if c8_site is not None and c8_site.primed:
            c8 = (torch.empty(x2.shape[0], w._q8.shape[0], device=x2.device, dtype=c8_site.torch_dtype), c8_site)
        y = K.gemm(x8, w._q8, bias=_bias(b), alpha_dev=(w._xsite.inv_scale, w._wsite.inv_scale), c8=c8, write_c=not (c8_only and c8 is not None), **kw)
        return (y, c8[0] if c8 is not None else None) if want8 else y
    assert not c8_only
    y = K.gemm(x2, w._lp, bias=_bias(b), **kw)
    return (y, None) if want8 else y


def lin_dx(dy2, w, fp8=False, g8=None, c8_site=None, want8=False, c8_only=False, c3_consumer=None, c3_bias=None, **kw):
    """dX = dY @ W.  bf16: NT against the K-contiguous [in,out] copy; f32: the generic kernel reads W as [K,N].
    fp8: dY in e5m2 under the Linear's gradient site against the e4m3 [in, out] weight copy."""
    _epi_bwd(kw)
    if fp8 == X3 and _is_x3(g8):              # G-form copy of dY against the X-form split of W^T ([in, 3 out], K-contiguous)
        w3t = _w_split(w, True)
        K.f32_split_launches += 1
        if want8 and c3_consumer is not None and fuse_x3_copies and K.NS == 3 and _x3_lin_ok(g8.shape[0], c3_consumer) \
                and K.gemm256_takes(g8.shape[0], w3t.shape[0], w3t.shape[1]):
            # the gradient this product writes is read only through its G-form copy (dX and weight-gradient product of the consumer):
            # written by the epilogue, no f32 tensor; the consumer's bias gradient = column sums of the copy's hi and lo segments
            _, y3 = K.gemm(g8, w3t, out_dtype=torch.float32, c3_form=K.FORM_G, write_c=False, **kw)
            if c3_bias is not None and not _x3_asum_ok(g8.shape[0], c3_consumer):      # (else: inside the consumer's weight-gradient launch, _lin_dw)
                n = w3t.shape[0]
                K.colsum(y3[:, :n], c3_bias._g.reshape(-1))
                K.colsum(y3[:, n:2 * n], c3_bias._g.reshape(-1))
                _mark(c3_bias)
                y3._bias_done = True
            return None, y3
        y = K.gemm(g8, w3t, out_dtype=torch.float32, **kw)
        return (y, None) if want8 else y
    if fp8 and fp8 != X3 and getattr(w, '_q8t', None) is not None and _fp8_ok(dy2, w._q8t):
        if g8 is None:
            g8 = w._gsite.quantize(dy2)
        c8 = None
        if c8_site is not None and c8_site.primed:
            c8 = (torch.empty(dy2.shape[0], w._q8t.shape[0], device=dy2.device, dtype=c8_site.torch_dtype), c8_site)
        y = K.gemm(g8, w._q8t, alpha_dev=(w._gsite.inv_scale, w._wsite.inv_scale), c8=c8, write_c=not (c8_only and c8 is not None), **kw)
        return (y, c8[0] if c8 is not None else None) if want8 else y
    assert not c8_only
    y = K.gemm(dy2, w._lpt, **kw) if w._lpt is not None else K.gemm(dy2, w._lp, trans_b=True, **kw)
    return (y, None) if want8 else y


# Weight-gradient launches (dW GEMM + bias column sums) run on a second HIP stream beside the dX chain of the backward pass:
# a dX GEMM and the dW GEMM of the same layer are independent, and each persistent GEMM ends with CUs idling through its last
# partial round of tiles (6-7 % of a 1731-tile launch) - the other stream's workgroups fill those CUs.  Measured on the batch-256
# step: 155.0 -> 150.8 ms.  LECCR_DW_STREAM=0 keeps everything on one stream.
wgrad_overlap = os.environ.get('LECCR_DW_STREAM', '1') != '0'
wgrad_overlap_f32 = os.environ.get('LECCR_DW_STREAM_F32', '0') != '0'      # .. for the fp32 path (see lin_dw)
wgrad_inline_under_pressure = os.environ.get('LECCR_DW_INLINE_PRESSURE', '1') != '0'   # .. and no side stream for a pass that starts under memory pressure
_pass_inline = None          # this backward pass runs its weight-gradient launches on the main stream (decided at its first one)
fuse_bias_grad = os.environ.get('LECCR_FUSE_BIAS_GRAD', '1') != '0'      # bias gradients inside the dW GEMM (0: separate colsum launches)
# The text towers (caption BERT forward, text BERT forward + backward: small-M GEMMs that leave CUs idle) run on a third stream
# beside the image tower (150.8 -> 148.7 ms on the batch-256 step).  LECCR_TEXT_STREAM=0 keeps them on the main stream.
text_overlap = os.environ.get('LECCR_TEXT_STREAM', '1') != '0'
wgrad_lag = int(os.environ.get('LECCR_DW_LAG', '12'))      # host run-ahead over the weight-gradient stream, in launches (0: unbounded)
wgrad_lag_pressure = float(os.environ.get('LECCR_DW_LAG_PRESSURE', '0.35'))  # .. applied when the backward pass starts with this share of the HBM allocated
wgrad_lag_reserved = float(os.environ.get('LECCR_DW_LAG_RESERVED', '0.75'))   # .. or with this share of the HBM held by the allocator's pool
_wgrad_events = collections.deque()
_wgrad_throttle = False
_bwd_done = None             # recorded when a backward pass (side stream included) has been queued completely
_wgrad_stream = None
_text_stream = None
_wgrad_join_queued = False


def begin_step():
    """Called at the start of every model forward: a backward pass that raised leaves its end-of-pass join unqueued-but-flagged;
    the next pass must queue its own."""
    global _wgrad_join_queued, _g8_slot, _pass_inline
    _pass_inline = None
    _wgrad_join_queued = False
    _g8_slot = None
    _wgrad_events.clear()


def text_stream(device):
    """The stream the text towers run on (None: on the current stream)."""
    global _text_stream
    if not (text_overlap and torch.device(device).type == 'cuda'):
        return None
    if _text_stream is None:
        _text_stream = torch.cuda.Stream(device=device)
    return _text_stream


_caption_stream = None


def caption_stream(device):
    """A further stream for the caption-BERT forward (no gradient, so nothing of the backward pass runs on it), or None
    (LECCR_CAP_STREAM=0): the two BERT forwards then overlap each other as well as the image tower (+1.3 % on the step)."""
    global _caption_stream
    if os.environ.get('LECCR_CAP_STREAM', '1') == '0' or torch.device(device).type != 'cuda':
        return None
    if _caption_stream is None:
        _caption_stream = torch.cuda.Stream(device=device)
    return _caption_stream


def wgrad_stream(device):
    """The weight-gradient stream (created at the first weight-gradient launch, or ahead of it by bind_streams)."""
    global _wgrad_stream
    if _wgrad_stream is None:
        _wgrad_stream = torch.cuda.Stream(device=device)
    return _wgrad_stream


_update_stream = None


def update_stream(device):
    """The stream of the early region's fused AdamW update (optim.ArenaAdamW._early_update)."""
    global _update_stream
    if _update_stream is None:
        cap = caption_stream(device) if os.environ.get('LECCR_UPDATE_STREAM', 'own') == 'caption' else None
        _update_stream = cap if cap is not None else torch.cuda.Stream(device=device)
    return _update_stream


def bind_streams(device):
    """HIP binds a stream to one of GPU_MAX_HW_QUEUES (4, and 5 or more is slower: DESIGN.md 5a) hardware queues at its FIRST launch, the
    least-shared queue first; streams on one queue run in order.  A single process finds main / caption / text / weight-gradient on a
    queue each (the update stream shares the weight-gradient one) simply by the order its first step uses them.  A process that joins a
    process group first does not: RCCL and c10d launch on half a dozen streams of their own at communicator creation, and the step's
    streams then land wherever is left - measured in a one-rank world: the caption tower on the MAIN stream's queue, +2.1 ms per step
    (profiles/r05_stream_queue_binding.txt).  Called by both init_distributed_mode paths BEFORE init_process_group: one tiny launch
    on each stream, in the order of a step."""
    device = torch.device(device)
    if device.type != 'cuda':
        return
    torch.zeros(1, device=device)
    for s in (caption_stream(device), text_stream(device), wgrad_stream(device) if wgrad_overlap else None, update_stream(device)):
        if s is not None:
            with torch.cuda.stream(s):
                torch.zeros(1, device=device)
    torch.cuda.synchronize(device)


def side_streams():
    return [s for s in (_wgrad_stream, _text_stream) if s is not None]


def join_wgrad_stream(stream=None):
    """`stream` (default: the current one) waits for everything issued so far on the side streams (weight-gradient launches,
    text-tower backward) - before the gradients are read."""
    streams = side_streams()
    if not streams:
        return
    cur = stream if stream is not None else torch.cuda.current_stream()
    for s in streams:
        cur.wait_stream(s)


def lin_dw(dy2, x2, w, b=None, scale=1.0, dy_bias=None, g8=None, x8=None, bias_done=False):
    """dW += scale * dY^T X (f32, into the gradient arena) and db += scale * colsum(dY).
    dy_bias: the tensor whose column sums are the bias gradient when it is not dy2 itself (an f32 head's incoming gradient,
    of which dy2 is the bf16 GEMM operand: the bias sum cancels heavily and is taken from the unrounded values).
    g8, x8 (fp8 mode): the e5m2 copy of dy2 and the e4m3 copy of x2 under w's gradient / activation sites - the product then runs on
    the fp8 weight-gradient kernel, the bias sums inside the same launch from the e5m2 copy (fp8_bias_asum; else a pass over the bf16 dy2);
    x2 may then hold only the tail rows (dw_keep).
    dy2 None (fp8_only_ok): only the fp8 copy of the gradient exists - no tail rows, and the bias gradient was summed by the
    producing GEMM's epilogue (bias_done)."""
    global _wgrad_stream, _wgrad_join_queued, _wgrad_throttle
    if _is_x3(g8) and _is_x3(x8) and g8.shape[0] == x8.shape[0]:
        pass            # mode X3: both split copies at hand - the product reads only them
    elif _is_x3(g8) or _is_x3(x8):
        if dy2 is None or x2.shape[0] != dy2.shape[0]:
            raise RuntimeError('lin_dw: a split copy without its partner and without the f32 operands')
        if _is_x3(g8):          # the copy is dropped, not what its producer already did: bias sums taken in its split pass are not taken again
            bias_done = bias_done or getattr(g8, '_bias_done', False)
        g8 = x8 = None
    elif not (fp8_dw and g8 is not None and x8 is not None and _fp8_dw_ok(g8, x8)):
        if dy2 is None or x2.shape[0] != dy2.shape[0]:
            raise RuntimeError('lin_dw: only the fp8 copy of an operand was kept (dw_keep / fp8_only_ok) but the fp8 weight-gradient product '
                               'is not available for dY8 %s / X8 %s' % (None if g8 is None else tuple(g8.shape), None if x8 is None else tuple(x8.shape)))
        g8 = x8 = None
    if _is_x3(g8):
        bias_done = bias_done or getattr(g8, '_bias_done', False)
    elif dy2 is None:
        assert g8.shape[0] % 128 == 0 and (b is None or bias_done or (fp8_bias_asum and fuse_bias_grad and scale == 1.0 and dy_bias is None))
    ref = dy2 if dy2 is not None else g8
    # fp32 path: the weight-gradient products stay on the main stream.  Their launches are long (reduction over 3 x rows) persistent
    # kernels that hold every CU, and the fp32 step's main chain is full of short HBM-bound launches (split passes, LayerNorm) that then
    # queue behind them: same-box A/B 397.0 ms with the side stream, 374.8 without (LECCR_DW_STREAM_F32=1 turns it back on)
    f32_path = _is_x3(g8) or (dy2 is not None and dy2.dtype == torch.float32)
    if not (wgrad_overlap and ref.is_cuda) or (f32_path and not wgrad_overlap_f32):
        return _lin_dw(dy2, x2, w, b, scale, dy_bias, g8, x8, bias_done)
    # Under memory pressure (the config-E geometry: 150-175 of 288 GB allocated when the backward pass starts) the side stream needs the
    # host throttle below to keep the allocator's overhang bounded - and then measures SLOWER than no side stream at all (config E fp8,
    # same box: 381.4 ms with it, 375.5 without; config B, no pressure: 130.6 with, 132.1 without).  Decided once per backward pass.
    global _pass_inline
    if _pass_inline is None:
        total = torch.cuda.get_device_properties(ref.device).total_memory
        _pass_inline = wgrad_inline_under_pressure and (torch.cuda.memory_allocated(ref.device) > wgrad_lag_pressure * total
                                                        or torch.cuda.memory_reserved(ref.device) > wgrad_lag_reserved * total)
    if _pass_inline:
        return _lin_dw(dy2, x2, w, b, scale, dy_bias, g8, x8, bias_done)
    main = torch.cuda.current_stream()
    side = wgrad_stream(ref.device)
    side.wait_stream(main)                      # dY and X were produced on the main stream
    with torch.cuda.stream(side):
        _lin_dw(dy2, x2, w, b, scale, dy_bias, g8, x8, bias_done)
    if dy2 is not None:
        dy2.record_stream(side)                 # the caching allocator must not recycle them under the side stream
    x2.record_stream(side)
    if g8 is not None:
        g8.record_stream(side)
        x8.record_stream(side)
    if wgrad_lag > 0 and _wgrad_throttle:
        # the caching allocator hands a block touched by the side stream back only once that stream has passed the point of the free,
        # and it looks at those events when the HOST allocates: a host that queues the whole backward pass ahead of the device therefore
        # holds every block's temporaries at once (config E: 5 GB per ViT-L block, reserved memory 286 of 288 GB, allocation retries).
        # Keeping the host at most `wgrad_lag` weight-gradient launches (3 blocks) ahead of the side stream bounds that overhang
        # (config E: reserved 286 -> 163 GB).  Only under memory pressure (wgrad_lag_pressure): on the batch-256 ViT-B/16 step, where the
        # overhang fits (reserved 107 of 288 GB), the host waits cost 2.4 ms of 130.6.
        # Hysteresis: the host waits only once 2 x wgrad_lag launches are outstanding, then until wgrad_lag are left - one blocking
        # wait (and one wake-up latency) per wgrad_lag launches instead of one per launch.
        ev = torch.cuda.Event()
        ev.record(side)
        _wgrad_events.append(ev)
        if len(_wgrad_events) >= 2 * wgrad_lag:
            while len(_wgrad_events) > wgrad_lag + 1:
                _wgrad_events.popleft()
            _wgrad_events.popleft().synchronize()
    if dy_bias is not None:
        dy_bias.record_stream(side)
    if not _wgrad_join_queued:
        # First weight-gradient launch of a backward pass.  The host does not start queuing this pass before the device has finished the
        # PREVIOUS one (it still has this step's whole forward queued, so nothing idles): a loop that never reads a loss (bench.py's
        # timed region; the reference's loop reads loss.item() every step) lets the host run steps ahead, the blocks the side stream
        # touched come back to the allocator only when the device gets there, and every step ahead allocates its temporaries anew -
        # the pool grew from 60 to 150-217 GB over 8-16 steps of the batch-256 step (and into allocation retries beyond).
        global _bwd_done
        if _bwd_done is not None:
            _bwd_done.synchronize()
            _bwd_done = None
        # (.. also the moment to look at the memory pressure: every saved activation is still held)
        # .. or the allocator's pool has grown past three quarters of the HBM (it never shrinks by itself)
        total = torch.cuda.get_device_properties(ref.device).total_memory
        _wgrad_throttle = wgrad_lag > 0 and (torch.cuda.memory_allocated(ref.device) > wgrad_lag_pressure * total
                                             or torch.cuda.memory_reserved(ref.device) > wgrad_lag_reserved * total)
        # once per backward pass: when the autograd engine has run every node, the main stream waits for the side stream,
        # so whatever follows backward() (optimizer, gradient all-reduce, a test reading .grad) sees complete gradients
        def _join():                            # (final callbacks run on the stream that called backward())
            global _wgrad_join_queued, _bwd_done
            _wgrad_join_queued = False
            join_wgrad_stream()
            _bwd_done = torch.cuda.Event()
            _bwd_done.record()
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join)
            _wgrad_join_queued = True
        except RuntimeError:                    # not inside a backward pass (a direct call): join right away
            main.wait_stream(side)


def _fp8_dw_ok(g8, x8):
    """leccr_gemm's fp8 weight-gradient form: whole 256 x 256 output tiles, >= 2048 rows (whole 128-row K-tiles; a shorter tail goes
    to the bf16 kernel), 16-byte aligned rows."""
    return (g8.shape[0] >= 2048 and g8.shape[1] % 256 == 0 and x8.shape[1] % 256 == 0 and g8.stride(0) % 16 == 0 and x8.stride(0) % 16 == 0
            and g8.data_ptr() % 16 == 0 and x8.data_ptr() % 16 == 0)


def _lin_dw(dy2, x2, w, b, scale, dy_bias=None, g8=None, x8=None, bias_done=False):
    # the bias gradient colsum(dY) rides in the dW GEMM (leccr_gemm_params.asum: on the matrix cores beside the product in the
    # 256x256 kernel, a colsum pass inside leccr_gemm otherwise) unless it comes from another tensor or carries a scale
    if bias_done:           # summed by the GEMM epilogue that produced the gradient (fp8_only_ok)
        _mark(b)
        b = None
    fused = b is not None and dy_bias is None and scale == 1.0 and fuse_bias_grad
    if _is_x3(g8):
        # f32 operands as bf16 hi / lo splits: the [rows, 3 cols] copies read as [3 rows, cols], one TN launch over 3 x rows
        rows = g8.shape[0]
        K.f32_split_launches += 1
        # (the split copy's plain column sums would count hi twice: the 4-wave TN kernel's asum_x3 form skips the repeated segment; where that
        #  kernel does not take the shape, a colsum pass over the f32 gradient below)
        fused = fused and _x3_asum_ok(rows, w)
        assert fused or b is None or dy2 is not None or dy_bias is not None, 'lin_dw: the bias gradient of a copy-only dY needs the asum_x3 product'
        K.gemm(g8.view(K.NS * rows, g8.shape[1] // K.NS), x8.view(K.NS * rows, x8.shape[1] // K.NS), trans_a=True, trans_b=True, out=w._g,
               accumulate=True, alpha=scale, asum=b._g.reshape(-1) if fused else None, asum_x3=fused)
    elif g8 is not None:
        # fp8 operands: whole 128-row K-tiles on the scaled-MFMA kernel, the (< 128-row) tail and the bias sums from the bf16 tensors
        global fp8_dw_launches
        fp8_dw_launches += 1
        # (round 4: the bias gradient rides in the same launches - the column sums of the e5m2 copy, the values the weight gradient itself
        #  is built from, in place of a pass over the bf16 tensor; LECCR_FP8_BIAS_ASUM=0: that pass again)
        rows = g8.shape[0] // 128 * 128
        fused = fused and fp8_bias_asum
        bsum = b._g.reshape(-1) if fused else None
        K.gemm(g8[:rows], x8[:rows], trans_a=True, trans_b=True, out=w._g, accumulate=True, alpha=scale,
               alpha_dev=(w._gsite.inv_scale, w._xsite.inv_scale), asum=bsum)
        if rows < g8.shape[0]:
            xt = x2[rows:] if x2.shape[0] == g8.shape[0] else x2          # full tensor, or the tail rows dw_keep kept
            assert xt.shape[0] == g8.shape[0] - rows
            K.gemm(dy2[rows:], xt, trans_a=True, trans_b=True, out=w._g, accumulate=True, alpha=scale, asum=bsum)
    else:
        K.gemm(dy2, x2, trans_a=True, trans_b=True, out=w._g, accumulate=True, alpha=scale,
               asum=b._g.reshape(-1) if fused else None)
    _mark(w)
    if b is not None and fused:
        _mark(b)
    elif b is not None:
        src = dy2 if dy_bias is None else dy_bias
        if scale == 1.0:
            K.colsum(src, b._g.reshape(-1))
        else:
            tmp = torch.zeros(b._g.numel(), device=dy2.device, dtype=torch.float32)
            K.colsum(src, tmp)
            K.axpy(tmp, b._g.reshape(-1), scale)
        _mark(b)


def _mark(p):
    """Record that a kernel accumulated into this parameter's gradient (drives the optimizer's skip list)."""
    for m in (getattr(p, 'members', None) or (p,)):
        m._arena.touched[m._aidx] = True


def _rows(x):
    return x.reshape(-1, x.shape[-1])


class _Slice:
    """Row range [r0, r1) of a 2-D weight/bias holder (packed in_proj: q rows, kv rows)."""

    def __init__(self, h, r0, r1, cols_of_t=None):
        self._lp = h._lp[r0:r1]
        self._g = h._g[r0:r1]
        self.data = h.data[r0:r1]
        self._lpt = h._lpt[:, r0:r1] if getattr(h, '_lpt', None) is not None and h._lpt.dim() == 2 else None
        self._base = h

    @property
    def members(self):
        return getattr(self._base, 'members', (self._base,))


# --------------------------------------------------------------------------------------- Linear / LayerNorm / L2
class LinearFn(torch.autograd.Function):
    """nn.Linear on the last axis.  x may be a strided row view (e.g. the [CLS] rows x[:, 0, :])."""

    @staticmethod
    def forward(ctx, x, weight, bias, wgrad_scale, out_f32):
        x2 = x if x.dim() == 2 else _rows(x)
        # fp32 path, a Linear large enough for the split-operand kernels (the projections over all image tokens): one X-form copy of x
        # serves the forward and the weight-gradient product, one G-form copy of dy the dX and the weight-gradient product (each
        # product splitting its own operands made four passes of these two)
        x8 = x8_of(X3, x2, weight) if (share_x3_copies and x2.dtype == torch.float32) else None
        if x8 is not None:
            y = lin_fwd(x2, weight, bias, fp8=X3, x8=x8)
        else:
            y = lin_fwd(x2, weight, bias, out_dtype=torch.float32 if out_f32 else None)
        ctx.save_for_backward(x2, x8)
        ctx.w, ctx.b, ctx.shape, ctx.scale = weight, bias, x.shape, wgrad_scale
        return y.view(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, x8 = ctx.saved_tensors
        dy2 = dyf = _rows(dy.contiguous())
        if dy2.dtype != x2.dtype:               # f32 head output: the GEMM operands share the compute dtype
            dy2 = dy2.to(x2.dtype)
        g8 = None
        if x8 is not None:
            g8 = g8_of(X3, dy2, ctx.w, bias=ctx.b if ctx.scale == 1.0 else None)
        if g8 is not None:
            dx = lin_dx(dy2, ctx.w, fp8=X3, g8=g8).view(ctx.shape) if ctx.needs_input_grad[0] else None
            lin_dw(dy2, x2, ctx.w, ctx.b, ctx.scale, g8=g8, x8=x8)
            return dx, None, None, None, None
        dx = lin_dx(dy2, ctx.w).view(ctx.shape) if ctx.needs_input_grad[0] else None
        lin_dw(dy2, x2, ctx.w, ctx.b, ctx.scale, dy_bias=None if dyf is dy2 else dyf)
        return dx, None, None, None, None


def linear(x, weight, bias=None, wgrad_scale=1.0, out_f32=False):
    """out_f32: write the result in f32 even when the operands are bf16 (projection heads feeding the f32 losses)."""
    return LinearFn.apply(x, weight, bias, wgrad_scale, out_f32)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        y, mean, rstd, _ = K.layernorm_fwd(x, weight.data, bias.data, eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.w, ctx.b = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dx = K.layernorm_bwd(dy.contiguous(), x, ctx.w.data, mean, rstd, ctx.w._g, ctx.b._g)
        _mark(ctx.w)
        _mark(ctx.b)
        return dx, None, None, None


def layer_norm(x, weight, bias, eps):
    return LayerNormFn.apply(x, weight, bias, eps)


class L2NormFn(torch.autograd.Function):
    """F.normalize(x, dim=-1) (models/xvlm.py:245-256)."""

    @staticmethod
    def forward(ctx, x):
        y, inv = K.l2norm_fwd(x.contiguous())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return K.l2norm_bwd(dy.contiguous(), y, inv)


def l2_normalize(x):
    return L2NormFn.apply(x)


# --------------------------------------------------------------------------------------- CLIP ViT
class PatchEmbedFn(torch.autograd.Function):
    """conv1 (stride == kernel) as patchify + GEMM, then [CLS] + positional embedding (clip/model.py:224-228)."""

    @staticmethod
    def forward(ctx, image, conv_w, cls, pos, dtype):
        b, _, R, _ = image.shape
        d, _, p, _ = conv_w.shape
        w2d = getattr(conv_w, '_lp_pad', None)          # zero-padded copy when 3*p*p is not a multiple of 64 (ViT-L/14)
        if w2d is None:
            w2d = conv_w._lp2d
        Kp = w2d.shape[1]
        patches = K.patchify(image.contiguous(), p, Kp, dtype)
        po = K.gemm(patches, w2d)
        tok = K.vit_assemble_fwd(po, cls.data, pos.data, b)
        ctx.save_for_backward(patches)
        ctx.p = (conv_w, cls, pos)
        return tok

    @staticmethod
    def backward(ctx, dtok):
        (patches,) = ctx.saved_tensors
        conv_w, cls, pos = ctx.p
        dpo = K.vit_assemble_bwd(dtok.contiguous(), cls._g, pos._g)
        _mark(cls)
        _mark(pos)
        if patches.shape[1] == conv_w._g2d.shape[1]:
            K.gemm(dpo, patches, trans_a=True, trans_b=True, out=conv_w._g2d, accumulate=True)
        else:                                       # padded patch rows: the padding columns' gradient is dropped
            gpad = torch.zeros(conv_w._g2d.shape[0], patches.shape[1], device=dpo.device, dtype=torch.float32)
            K.gemm(dpo, patches, trans_a=True, trans_b=True, out=gpad, accumulate=True)
            conv_w._g2d.add_(gpad[:, :conv_w._g2d.shape[1]])
        _mark(conv_w)
        return None, None, None, None, None


class ViTBlockFn(torch.autograd.Function):
    """ResidualAttentionBlock (clip/model.py:171-192): x + MHA(ln_1(x)); then + c_proj(QuickGELU(c_fc(ln_2(.))))."""

    @staticmethod
    def forward(ctx, x, blk, *params):
        b, N, d = x.shape
        M = b * N
        # (fp8: the LayerNorm passes also write the e4m3 copy their consumer multiplies - no quantise pass over h1 / h2)
        # (mode X3: h1 / h2 are read only as split copies - the LayerNorm pass writes the copy and no f32 tensor)
        x3 = blk.fp8 == X3
        h1, mean1, rstd1, _, h1_8 = _ln_fwd(x, blk.ln_1.weight.data, blk.ln_1.bias.data, 1e-5,
                                            site=_q8_site(blk.fp8, M, blk.attn.in_proj_weight, '_xsite'),
                                            x3_for=blk.attn.in_proj_weight if x3 else None, x3_only=True,
                                            q8_only=bool(blk.fp8) and not x3 and fp8_only_ok(M, blk.attn.in_proj_weight, False))
        h1r = _rows(h1) if h1 is not None else (None if x3 else h1_8)          # (fp8: the copy stands in for the tensor's shape)
        h1_8 = x8_of(blk.fp8, h1r, blk.attn.in_proj_weight, h1_8)
        qkv = lin_fwd(h1r, blk.attn.in_proj_weight, blk.attn.in_proj_bias, fp8=blk.fp8, x8=h1_8).view(b, N, 3 * d)
        # (fp8: the attention kernel writes the e4m3 copy of o that out_proj multiplies, the backward the e5m2 copy of dqkv for in_proj)
        o_site = _q8_site(blk.fp8, M, blk.attn.out_proj.weight, '_xsite')
        if getattr(blk, 'causal', False):              # CLIP's text transformer (clip/model.py:286, 328-334): key j <= query i
            o, lse = K.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], blk.heads, causal=True)
            o_8 = None
        elif o_site is not None and o_site.primed:
            o, lse, o_8 = K.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], blk.heads, q8_site=o_site)
        elif x3 and attn_x3_copies and fuse_x3_copies and _x3_lin_ok(M, blk.attn.out_proj.weight) and _x3_lin_ok(M, blk.attn.in_proj_weight) \
                and K.attn_s3_takes(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], blk.heads):
            # mode X3: o is read only as out_proj's X-form copy (forward and weight-gradient product) and by the backward's delta = dO . O:
            # the attention kernel writes that copy and no f32 tensor (the backward reads o back as hi + lo)
            o_8, lse = K.attn_fwd_s3(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], blk.heads)
            o = None
        else:
            o, lse = K.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], blk.heads)
            o_8 = x8_of(blk.fp8, _rows(o), blk.attn.out_proj.weight)
        x1 = lin_fwd(_rows(o) if o is not None else None, blk.attn.out_proj.weight, blk.attn.out_proj.bias, fp8=blk.fp8, x8=o_8,
                     residual=_rows(x)).view(b, N, d)
        h2, mean2, rstd2, _, h2_8 = _ln_fwd(x1, blk.ln_2.weight.data, blk.ln_2.bias.data, 1e-5,
                                            site=_q8_site(blk.fp8, M, blk.mlp.c_fc.weight, '_xsite'),
                                            x3_for=blk.mlp.c_fc.weight if x3 else None, x3_only=True,
                                            q8_only=bool(blk.fp8) and not x3 and fp8_only_ok(M, blk.mlp.c_fc.weight, False))
        h2r = _rows(h2) if h2 is not None else (None if x3 else h2_8)
        pre = torch.empty(M, 4 * d, device=x.device, dtype=x.dtype)
        h2_8 = x8_of(blk.fp8, h2r, blk.mlp.c_fc.weight, h2_8)
        # (fp8: the activation's e4m3 copy for c_proj comes out of c_fc's epilogue - no quantise pass over the 4d-wide tensor)
        # (.. and when c_proj's forward and weight-gradient products both read that copy, the bf16 activation is not written at all)
        act_8only = blk.fp8 and h2_8 is not None and fp8_only_ok(M, blk.mlp.c_proj.weight, False)
        act, act8 = lin_fwd(h2r, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, fp8=blk.fp8, x8=h2_8, epilogue=EPI_QUICKGELU, aux_out=pre,
                            c8_site=_site(blk.mlp.c_proj.weight, '_xsite'), want8=True, c8_only=act_8only, c3_consumer=blk.mlp.c_proj.weight)
        if act is None:
            act = pre[:0]                                   # (the tail rows dw_keep would keep: none)
            x2 = lin_fwd(act8, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, fp8=blk.fp8, x8=act8, residual=_rows(x1)).view(b, N, d)
        else:
            act8 = x8_of(blk.fp8, act, blk.mlp.c_proj.weight, act8)
            x2 = lin_fwd(act, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, fp8=blk.fp8, x8=act8, residual=_rows(x1)).view(b, N, d)
        # fp8: the e4m3 copies the forward products multiplied are the weight-gradient products' X operands, and the bf16 h1 / h2 / act
        # are kept for nothing else (dw_keep)
        ctx.save_for_backward(x, dw_keep(h1r, h1_8, blk.attn.in_proj_weight), mean1, rstd1, qkv, o, lse, x1,
                              dw_keep(h2r, h2_8, blk.mlp.c_fc.weight), mean2, rstd2, pre,
                              act if act.shape[0] == 0 else dw_keep(act, act8, blk.mlp.c_proj.weight),
                              *((h1_8, o_8, h2_8, act8) if (fp8_dw or o is None or blk.fp8 == X3) else (None,) * 4))
        ctx.blk = blk
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x, h1, mean1, rstd1, qkv, o, lse, x1, h2, mean2, rstd2, pre, act, h1_8, o_8, h2_8, act8 = ctx.saved_tensors
        blk = ctx.blk
        _fire(blk.ln_1.weight, 'on_vit_block_backward', blk)
        b, N, d = x.shape
        dx2 = dx2.contiguous()
        g2 = _rows(dx2)
        g2_8 = _g8_take(g2) if blk.fp8 else None          # written by the next block's LayerNorm-1 backward (fp8 mode)
        g2_8 = g8_of(blk.fp8, g2, blk.mlp.c_proj.weight, g2_8, bias=blk.mlp.c_proj.bias)
        # (fp8: when c_fc's dX and weight-gradient products both read dpre's e5m2 copy, the bf16 dpre is not written; its column sums -
        #  c_fc's bias gradient - then come out of the same epilogue)
        dpre_8only = blk.fp8 and g2_8 is not None and h2_8 is not None and fp8_only_ok(g2.shape[0], blk.mlp.c_fc.weight, True)
        dpre, dpre8 = lin_dx(g2, blk.mlp.c_proj.weight, fp8=blk.fp8, g8=g2_8, epilogue=EPI_DQUICKGELU, aux=pre,
                             c8_site=_site(blk.mlp.c_fc.weight, '_gsite'), want8=True, c8_only=dpre_8only,
                             c3_consumer=blk.mlp.c_fc.weight, c3_bias=blk.mlp.c_fc.bias,
                             **(dict(colsum=blk.mlp.c_fc.bias._g.reshape(-1)) if dpre_8only else {}))
        # bias gradients of c_proj / out_proj = column sums of dx2 / dx1.  They ride the weight-gradient stream as 40 us HBM-bound
        # passes beside the dX chain: folded into the LayerNorm backward launches (whose `dres` operands those tensors are) they
        # cost that kernel a fifth of its occupancy - 253 us against 201 us per launch on the main chain
        lin_dw(g2, act, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, g8=g2_8, x8=act8)
        if dpre is not None:
            dpre8 = g8_of(blk.fp8, dpre, blk.mlp.c_fc.weight, dpre8, bias=blk.mlp.c_fc.bias)
        dh2 = lin_dx(dpre if dpre is not None else dpre8, blk.mlp.c_fc.weight, fp8=blk.fp8, g8=dpre8)
        # c_fc's bias gradient = column sums of dpre: an HBM-bound pass, issued with the dW GEMM on the weight-gradient stream
        # where it runs beside the (MFMA-bound) dX chain instead of in front of it
        lin_dw(dpre, h2, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, g8=dpre8, x8=h2_8,
               bias_done=getattr(dpre8, '_bias_done', False) if _is_x3(dpre8) else dpre is None)
        dx1, _, g1_8 = _ln_bwd(dh2.view(b, N, d), x1, blk.ln_2.weight.data, mean2, rstd2, blk.ln_2.weight._g, blk.ln_2.bias._g,
                               site=_q8_site(blk.fp8, b * N, blk.attn.out_proj.weight, '_gsite'), dres=dx2,
                               x3_for=blk.attn.out_proj.weight if blk.fp8 == X3 else None)
        g1 = _rows(dx1)
        g1_8 = g8_of(blk.fp8, g1, blk.attn.out_proj.weight, g1_8, bias=blk.attn.out_proj.bias)
        do = lin_dx(g1, blk.attn.out_proj.weight, fp8=blk.fp8, g8=g1_8).view(b, N, d)
        if o is None:
            # mode X3 with the attention kernels' own operand copies (forward above): o exists as o_8 only, and the gradient of the packed qkv
            # is written as in_proj's G-form copy only; in_proj's bias gradient is summed by the same kernels from the tiles they stage
            lin_dw(g1, o_8.new_empty((0, d), dtype=torch.float32), blk.attn.out_proj.weight, blk.attn.out_proj.bias, g8=g1_8, x8=o_8)
            dqkv3 = K.attn_bwd_s3(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], o_8, do, lse, blk.heads,
                                  bias_sum=blk.attn.in_proj_bias._g.reshape(-1))
            _mark(blk.attn.in_proj_bias)
            dqkv3._bias_done = True
            dh1 = lin_dx(None, blk.attn.in_proj_weight, fp8=blk.fp8, g8=dqkv3)
            lin_dw(None, h1, blk.attn.in_proj_weight, blk.attn.in_proj_bias, g8=dqkv3, x8=h1_8)
            return ViTBlockFn._finish_backward(ctx, blk, b, N, d, dh1, x, mean1, rstd1, dx1)
        lin_dw(g1, _rows(o), blk.attn.out_proj.weight, blk.attn.out_proj.bias, g8=g1_8, x8=o_8)
        g_site = _q8_site(blk.fp8, b * N, blk.attn.in_proj_weight, '_gsite')
        dqkv8 = None
        if g_site is not None and g_site.primed and not getattr(blk, 'causal', False):
            dqkv8 = torch.empty(qkv.shape, device=qkv.device, dtype=g_site.torch_dtype)
        # fp8: when in_proj's dX and weight-gradient products both read the e5m2 copy (and the bias sums ride in the latter), nothing reads
        # the bf16 gradient of the packed qkv: the attention kernels write the copy alone - a third of their output bytes
        only8 = (dqkv8 is not None and attn_q8_only and fp8_bias_asum and fuse_bias_grad and d // blk.heads == 64 and h1_8 is not None
                 and fp8_only_ok(b * N, blk.attn.in_proj_weight, True))
        dqkv = None if only8 else torch.empty_like(qkv)
        K.attn_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], o, do, lse, blk.heads,
                   dq=None if only8 else dqkv[..., :d], dk=None if only8 else dqkv[..., d:2 * d], dv=None if only8 else dqkv[..., 2 * d:],
                   causal=getattr(blk, 'causal', False), write_bf16=not only8,
                   q8=None if dqkv8 is None else (dqkv8[..., :d], dqkv8[..., d:2 * d], dqkv8[..., 2 * d:], g_site))
        if only8:
            dqkv8 = _rows(dqkv8)
            dh1 = lin_dx(dqkv8, blk.attn.in_proj_weight, fp8=blk.fp8, g8=dqkv8)
            lin_dw(None, h1, blk.attn.in_proj_weight, blk.attn.in_proj_bias, g8=dqkv8, x8=h1_8)
        else:
            dqkv8 = g8_of(blk.fp8, _rows(dqkv), blk.attn.in_proj_weight, None if dqkv8 is None else _rows(dqkv8), bias=blk.attn.in_proj_bias)
            dh1 = lin_dx(_rows(dqkv), blk.attn.in_proj_weight, fp8=blk.fp8, g8=dqkv8)
            lin_dw(_rows(dqkv), h1, blk.attn.in_proj_weight, blk.attn.in_proj_bias, g8=dqkv8, x8=h1_8)
        return ViTBlockFn._finish_backward(ctx, blk, b, N, d, dh1, x, mean1, rstd1, dx1)

    @staticmethod
    def _finish_backward(ctx, blk, b, N, d, dh1, x, mean1, rstd1, dx1):
        # (fp8: dx is the gradient the previous block's c_proj dX product consumes: its e5m2 copy goes with it)
        prev = getattr(blk, '_dx_consumer', (None,))[0]
        dx, _, dx_8 = _ln_bwd(dh1.view(b, N, d), x, blk.ln_1.weight.data, mean1, rstd1, blk.ln_1.weight._g, blk.ln_1.bias._g,
                              site=_q8_site(blk.fp8 and prev is not None, b * N, prev, '_gsite'), dres=dx1,
                              x3_for=prev if blk.fp8 == X3 else None)
        if blk.fp8:
            _g8_give(_rows(dx), dx_8)
        for p in (blk.ln_1.weight, blk.ln_1.bias, blk.ln_2.weight, blk.ln_2.bias):
            _mark(p)
        return (dx, None) + (None,) * 12


class LnProjFn(torch.autograd.Function):
    """hidden = ln_post(x) @ proj on ALL tokens (clip/model.py:340-346); proj is [d, vw] (no bias)."""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, proj):
        b, N, d = x.shape
        h, mean, rstd, _ = K.layernorm_fwd(x, ln_w.data, ln_b.data, 1e-5)
        # proj [d, vw]: B(n, k) = proj[k, n]  -> the [vw, d] K-contiguous copy is proj._lpt
        h3 = None
        if share_x3_copies and proj._lpt is None and K._split_plan(_rows(h), proj._lp, False, True, None, False, None) is not None and K.NS == 3:
            h3 = K.split3(_rows(h), K.FORM_X)          # fp32 path: one copy of h for this product and for dproj = H^T dY
        if h3 is not None:
            y = K.gemm(_rows(h), proj._lp, trans_b=True, a3=(h3, K.FORM_X))
        else:
            y = K.gemm(_rows(h), proj._lpt) if proj._lpt is not None else K.gemm(_rows(h), proj._lp, trans_b=True)
        ctx.save_for_backward(x, h, mean, rstd, h3)
        ctx.p = (ln_w, ln_b, proj)
        return y.view(b, N, -1)

    @staticmethod
    def backward(ctx, dy):
        x, h, mean, rstd, h3 = ctx.saved_tensors
        ln_w, ln_b, proj = ctx.p
        _fire(ln_w, 'on_late_backward_start')
        _fire(ln_w, 'on_early_grads_final')
        b, N, d = x.shape
        dy2 = _rows(dy.contiguous())
        if h3 is not None and dy2.dtype == torch.float32 and K._split_plan(dy2, proj._lp, False, False, None, False, None) is not None \
                and K._split_plan(_rows(h), dy2, True, True, proj._g, True, None) is not None:
            dy3 = K.split3(dy2, K.FORM_G)              # .. and one copy of dY for dH and dproj
            dh = K.gemm(dy2, proj._lp, a3=(dy3, K.FORM_G))
            K.gemm(_rows(h), dy2, trans_a=True, trans_b=True, out=proj._g, accumulate=True, a3=(h3, K.FORM_X), b3=(dy3, K.FORM_G))
        else:
            dh = K.gemm(dy2, proj._lp)                                    # dH = dY @ proj^T : B(n=d, k=vw) = proj[n, k]
            K.gemm(_rows(h), dy2, trans_a=True, trans_b=True, out=proj._g, accumulate=True)   # dproj[d,vw] = H^T dY
        _mark(proj)
        dx = K.layernorm_bwd(dh.view(b, N, d), x, ln_w.data, mean, rstd, ln_w._g, ln_b._g)
        _mark(ln_w)
        _mark(ln_b)
        return dx, None, None, None


# --------------------------------------------------------------------------------------- BERT
class BertEmbedFn(torch.autograd.Function):
    """BertEmbeddings: word + position + token_type(0) -> LayerNorm(1e-12) (HF BertModel, called at xvlm.py:238)."""

    @staticmethod
    def forward(ctx, ids, emb, dtype, *params):
        e = K.bert_embed_fwd(ids.contiguous(), emb.word_embeddings.weight.data, emb.position_embeddings.weight.data,
                             emb.token_type_embeddings.weight.data, dtype)
        y, mean, rstd, _ = K.layernorm_fwd(e, emb.LayerNorm.weight.data, emb.LayerNorm.bias.data, emb.eps)
        ctx.drop = (_p(emb, 'dropout_p'), _next_seed())
        if ctx.drop[0] > 0:
            K.dropout(y, ctx.drop[0], ctx.drop[1], out=y)
        ctx.save_for_backward(ids, e, mean, rstd)
        ctx.emb = emb
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, e, mean, rstd = ctx.saved_tensors
        emb = ctx.emb
        dy = dy.contiguous()
        if ctx.drop[0] > 0:
            dy = K.dropout(dy, ctx.drop[0], ctx.drop[1])
        de = K.layernorm_bwd(dy, e, emb.LayerNorm.weight.data, mean, rstd, emb.LayerNorm.weight._g,
                             emb.LayerNorm.bias._g)
        K.bert_embed_bwd(ids, de, emb.word_embeddings.weight._g, emb.position_embeddings.weight._g,
                         emb.token_type_embeddings.weight._g, emb.pad_id)
        for p in (emb.LayerNorm.weight, emb.LayerNorm.bias, emb.word_embeddings.weight,
                  emb.position_embeddings.weight, emb.token_type_embeddings.weight):
            _mark(p)
        return (None, None, None) + (None,) * 5


class BertLayerFn(torch.autograd.Function):
    """One post-LN BERT layer: packed QKV GEMM -> masked attention -> dense (+dropout) + residual -> LN -> GELU FFN
    (+dropout) + residual -> LN.  Dropout sites as in HF BertSelfAttention / BertSelfOutput / BertOutput."""

    @staticmethod
    def forward(ctx, x, kmask, layer, *params):
        b, L, d = x.shape
        ph, pa = _p(layer, 'hidden_dropout_p'), _p(layer, 'attn_dropout_p')
        seeds = (_next_seed(), _next_seed(), _next_seed())
        x_8 = x8_of(layer.fp8, _rows(x), layer._qkv_w)
        qkv = lin_fwd(_rows(x), layer._qkv_w, layer._qkv_b, fp8=layer.fp8, x8=x_8).view(b, L, 3 * d)
        ao = layer.attention.output
        o_site = _q8_site(layer.fp8, b * L, ao.dense.weight, '_xsite')          # fp8: o's e4m3 copy out of the attention kernel
        if o_site is not None and o_site.primed:
            o, lse, o_8 = K.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], layer.heads, kmask, pa, seeds[0], q8_site=o_site)
        else:
            o, lse = K.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], layer.heads, kmask, pa, seeds[0])
            o_8 = x8_of(layer.fp8, _rows(o), ao.dense.weight)
        x1_site = _q8_site(layer.fp8, b * L, layer.intermediate.dense.weight, '_xsite')      # fp8: x1's e4m3 copy out of its LayerNorm pass
        if ph > 0:      # dense -> dropout -> + residual (inside the LayerNorm kernel)
            a1 = lin_fwd(_rows(o), ao.dense.weight, ao.dense.bias, fp8=layer.fp8, x8=o_8)
            x1, mean1, rstd1, s1, x1_8 = _ln_fwd(a1.view(b, L, d), ao.LayerNorm.weight.data, ao.LayerNorm.bias.data, layer.eps,
                                                 site=x1_site, add=x, want_sum=True, drop=(ph, seeds[1]))
        else:
            s1 = lin_fwd(_rows(o), ao.dense.weight, ao.dense.bias, fp8=layer.fp8, x8=o_8, residual=_rows(x)).view(b, L, d)
            x1, mean1, rstd1, _, x1_8 = _ln_fwd(s1, ao.LayerNorm.weight.data, ao.LayerNorm.bias.data, layer.eps, site=x1_site)
        pre = torch.empty(b * L, layer.intermediate.dense.weight.shape[0], device=x.device, dtype=x.dtype)
        x1_8 = x8_of(layer.fp8, _rows(x1), layer.intermediate.dense.weight, x1_8)
        out = layer.output
        h_8only = layer.fp8 and x1_8 is not None and fp8_only_ok(b * L, out.dense.weight, False)       # as the ViT block's activation
        h, h8 = lin_fwd(_rows(x1), layer.intermediate.dense.weight, layer.intermediate.dense.bias, fp8=layer.fp8, x8=x1_8, epilogue=EPI_GELU,
                        aux_out=pre, c8_site=_site(out.dense.weight, '_xsite'), want8=True, c8_only=h_8only, c3_consumer=out.dense.weight)
        if h is None:
            h = h8                                          # the products below take the fp8 copy; no bf16 rows are kept
        else:
            h8 = x8_of(layer.fp8, h, out.dense.weight, h8)
        if ph > 0:
            a2 = lin_fwd(h, out.dense.weight, out.dense.bias, fp8=layer.fp8, x8=h8)
            x2, mean2, rstd2, s2 = K.layernorm_fwd(a2.view(b, L, d), out.LayerNorm.weight.data, out.LayerNorm.bias.data,
                                                   layer.eps, add=x1, want_sum=True, drop=(ph, seeds[2]))
        else:
            s2 = lin_fwd(h, out.dense.weight, out.dense.bias, fp8=layer.fp8, x8=h8, residual=_rows(x1)).view(b, L, d)
            x2, mean2, rstd2, _ = K.layernorm_fwd(s2, out.LayerNorm.weight.data, out.LayerNorm.bias.data, layer.eps)
        ctx.save_for_backward(dw_keep(_rows(x), x_8, layer._qkv_w), kmask, qkv, o, lse, s1, mean1, rstd1,
                              dw_keep(_rows(x1), x1_8, layer.intermediate.dense.weight), pre,
                              pre[:0] if h is h8 else dw_keep(h, h8, out.dense.weight), s2, mean2, rstd2,
                              *((x_8, o_8, x1_8, h8) if (fp8_dw or layer.fp8 == X3) else (None,) * 4))   # fp8 / X3: the weight-gradient products' X operands
        # (X3 keeps them whatever LECCR_FP8_DW says: dw_keep() has already dropped the f32 tensors these copies stand for - with the fp8 switch
        # off the fp32 step used to die in lin_dw: "a split copy without its partner")
        ctx.layer, ctx.drop = layer, (ph, pa, seeds)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        x, kmask, qkv, o, lse, s1, mean1, rstd1, x1, pre, h, s2, mean2, rstd2, x_8, o_8, x1_8, h8 = ctx.saved_tensors
        layer = ctx.layer
        ph, pa, seeds = ctx.drop
        ao, out = layer.attention.output, layer.output
        b, L, d = s1.shape
        # gradient of the sum: the residual path takes it as is (g2), the dense path through the dropout mask (gd2) - both
        # written by the LayerNorm backward pass
        # written by the LayerNorm backward pass (fp8: with gd2's e5m2 copy)
        g2, gd2, gd2_8 = _ln_bwd(dx2.contiguous(), s2, out.LayerNorm.weight.data, mean2, rstd2, out.LayerNorm.weight._g,
                                 out.LayerNorm.bias._g, site=_q8_site(layer.fp8, b * L, out.dense.weight, '_gsite'),
                                 drop=(ph, seeds[2]) if ph > 0 else None, x3_for=out.dense.weight if layer.fp8 == X3 else None)
        g2, gd2 = _rows(g2), _rows(gd2)
        gd2_8 = g8_of(layer.fp8, gd2, out.dense.weight, gd2_8, bias=out.dense.bias)
        inter = layer.intermediate.dense
        dpre_8only = layer.fp8 and gd2_8 is not None and x1_8 is not None and fp8_only_ok(gd2.shape[0], inter.weight, True)
        dpre, dpre8 = lin_dx(gd2, out.dense.weight, fp8=layer.fp8, g8=gd2_8, epilogue=EPI_DGELU, aux=pre,
                             c8_site=_site(inter.weight, '_gsite'), want8=True, c8_only=dpre_8only, c3_consumer=inter.weight, c3_bias=inter.bias,
                             **(dict(colsum=inter.bias._g.reshape(-1)) if dpre_8only else {}))
        lin_dw(gd2, h, out.dense.weight, out.dense.bias, g8=gd2_8, x8=h8)
        if dpre is not None:
            dpre8 = g8_of(layer.fp8, dpre, inter.weight, dpre8, bias=inter.bias)
        dx1 = lin_dx(dpre if dpre is not None else dpre8, inter.weight, fp8=layer.fp8, g8=dpre8, residual=g2)          # + residual path of s2 = .. + x1
        lin_dw(dpre, x1, inter.weight, inter.bias, g8=dpre8, x8=x1_8, bias_done=getattr(dpre8, '_bias_done', False) if _is_x3(dpre8) else dpre is None)
        g1, gd1, gd1_8 = _ln_bwd(dx1.view(b, L, d), s1, ao.LayerNorm.weight.data, mean1, rstd1, ao.LayerNorm.weight._g,
                                 ao.LayerNorm.bias._g, site=_q8_site(layer.fp8, b * L, ao.dense.weight, '_gsite'),
                                 drop=(ph, seeds[1]) if ph > 0 else None, x3_for=ao.dense.weight if layer.fp8 == X3 else None)
        g1, gd1 = _rows(g1), _rows(gd1)
        gd1_8 = g8_of(layer.fp8, gd1, ao.dense.weight, gd1_8, bias=ao.dense.bias)
        do = lin_dx(gd1, ao.dense.weight, fp8=layer.fp8, g8=gd1_8).view(b, L, d)
        lin_dw(gd1, _rows(o), ao.dense.weight, ao.dense.bias, g8=gd1_8, x8=o_8)
        dqkv = torch.empty_like(qkv)
        g_site = _q8_site(layer.fp8, b * L, layer._qkv_w, '_gsite')            # fp8: dqkv's e5m2 copy out of the attention backward
        dqkv8 = torch.empty(qkv.shape, device=qkv.device, dtype=g_site.torch_dtype) if (g_site is not None and g_site.primed) else None
        K.attn_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], o, do, lse, layer.heads, kmask,
                   dq=dqkv[..., :d], dk=dqkv[..., d:2 * d], dv=dqkv[..., 2 * d:], drop_p=pa, seed=seeds[0],
                   q8=None if dqkv8 is None else (dqkv8[..., :d], dqkv8[..., d:2 * d], dqkv8[..., 2 * d:], g_site))
        dqkv8 = g8_of(layer.fp8, _rows(dqkv), layer._qkv_w, None if dqkv8 is None else _rows(dqkv8), bias=layer._qkv_b)
        dx = lin_dx(_rows(dqkv), layer._qkv_w, fp8=layer.fp8, g8=dqkv8, residual=g1).view(b, L, d)       # + residual path of s1 = .. + x
        lin_dw(_rows(dqkv), x, layer._qkv_w, layer._qkv_b, g8=dqkv8, x8=x_8)
        for p in (out.LayerNorm.weight, out.LayerNorm.bias, ao.LayerNorm.weight, ao.LayerNorm.bias):
            _mark(p)
        return (dx, None, None) + (None,) * len(ctx.needs_input_grad[3:])


# --------------------------------------------------------------------------------------- caption interaction
class CrossAttnLayerFn(torch.autograd.Function):
    """CrossAttentionLayer / SelfAttentionLayer (models/attention.py:24-128), batch-major:
    a = MHA(tgt, memory, key_padding_mask, dropout p); y = norm2(tgt + dropout2(linear2(dropout(gelu(linear1(a)))))).
    (norm1 unused, a NOT residual-added before the FFN - attention.py:47-54.)"""

    @staticmethod
    def forward(ctx, tgt, memory, kmask, layer, *params):
        b, Lq, d = tgt.shape
        Lk = memory.shape[1]
        mha = layer.multihead_attn
        pd = _p(layer, 'dropout_p')
        seeds = (_next_seed(), _next_seed(), _next_seed())
        wq, wkv = _Slice(mha.in_proj_weight, 0, d), _Slice(mha.in_proj_weight, d, 3 * d)
        bq, bkv = _Slice(mha.in_proj_bias, 0, d), _Slice(mha.in_proj_bias, d, 3 * d)
        # fp32 path (m = X3): a Linear large enough for the split-operand kernels - the ones over all image tokens - makes ONE X-form copy of
        # its input for the forward and the weight-gradient product (and one G-form copy of its incoming gradient in the backward)
        m = X3 if (share_x3_copies and tgt.dtype == torch.float32) else False
        tq8 = x8_of(m, _rows(tgt), wq)
        q = lin_fwd(_rows(tgt), wq, bq, fp8=m, x8=tq8).view(b, Lq, d)
        mem8 = x8_of(m, _rows(memory), wkv)
        kv = lin_fwd(_rows(memory), wkv, bkv, fp8=m, x8=mem8).view(b, Lk, 2 * d)
        o, lse = K.attn_fwd(q, kv[..., :d], kv[..., d:], layer.heads, kmask, pd, seeds[0])
        o8 = x8_of(m, _rows(o), mha.out_proj.weight)
        a = lin_fwd(_rows(o), mha.out_proj.weight, mha.out_proj.bias, fp8=m, x8=o8)
        pre = torch.empty_like(a)
        a8 = x8_of(m, a, layer.linear1.weight)
        h = lin_fwd(a, layer.linear1.weight, layer.linear1.bias, fp8=m, x8=a8, epilogue=EPI_GELU, aux_out=pre)
        if pd > 0:
            K.dropout(h, pd, seeds[1], out=h)
            h8 = x8_of(m, h, layer.linear2.weight)
            l2 = lin_fwd(h, layer.linear2.weight, layer.linear2.bias, fp8=m, x8=h8)
            y, mean, rstd, s = K.layernorm_fwd(l2.view(b, Lq, d), layer.norm2.weight.data, layer.norm2.bias.data, 1e-5,
                                               add=tgt.contiguous(), want_sum=True, drop=(pd, seeds[2]))
        else:
            h8 = x8_of(m, h, layer.linear2.weight)
            s = lin_fwd(h, layer.linear2.weight, layer.linear2.bias, fp8=m, x8=h8, residual=_rows(tgt)).view(b, Lq, d)
            y, mean, rstd, _ = K.layernorm_fwd(s, layer.norm2.weight.data, layer.norm2.bias.data, 1e-5)
        # (a and h are read by nothing but their Linear's weight-gradient product: with a copy at hand the f32 tensors are not kept)
        ctx.save_for_backward(tgt, memory, kmask, q, kv, o, lse, dw_keep(a, a8, layer.linear1.weight), pre, dw_keep(h, h8, layer.linear2.weight),
                              s, mean, rstd, tq8, mem8, o8, a8, h8)
        ctx.layer, ctx.drop = layer, (pd, seeds)
        return y

    @staticmethod
    def backward(ctx, dy):
        tgt, memory, kmask, q, kv, o, lse, a, pre, h, s, mean, rstd, tq8, mem8, o8, a8, h8 = ctx.saved_tensors
        layer = ctx.layer
        mode = lambda x8: X3 if _is_x3(x8) else False          # (a Linear whose forward ran on its input's split copy)
        pd, seeds = ctx.drop
        mha = layer.multihead_attn
        b, Lq, d = tgt.shape
        Lk = memory.shape[1]
        wq, wkv = _Slice(mha.in_proj_weight, 0, d), _Slice(mha.in_proj_weight, d, 3 * d)
        bq, bkv = _Slice(mha.in_proj_bias, 0, d), _Slice(mha.in_proj_bias, d, 3 * d)
        ds = K.layernorm_bwd(dy.contiguous(), s, layer.norm2.weight.data, mean, rstd, layer.norm2.weight._g,
                             layer.norm2.bias._g, drop=(pd, seeds[2]) if pd > 0 else None)
        _mark(layer.norm2.weight)
        _mark(layer.norm2.bias)
        gs, gd = (_rows(ds[0]), _rows(ds[1])) if pd > 0 else (_rows(ds), _rows(ds))
        g8 = g8_of(mode(h8), gd, layer.linear2.weight, bias=layer.linear2.bias)
        dpre = lin_dx(gd, layer.linear2.weight, fp8=mode(h8), g8=g8, epilogue=EPI_DGELU, aux=pre)      # (g W2) * gelu'(pre) ...
        if pd > 0:
            K.dropout(dpre, pd, seeds[1], out=dpre)                               # ... * dropout mask (both elementwise)
        lin_dw(gd, h, layer.linear2.weight, layer.linear2.bias, g8=g8, x8=h8)
        g8 = g8_of(mode(a8), dpre, layer.linear1.weight, bias=layer.linear1.bias)
        da = lin_dx(dpre, layer.linear1.weight, fp8=mode(a8), g8=g8)
        lin_dw(dpre, a, layer.linear1.weight, layer.linear1.bias, g8=g8, x8=a8)
        g8 = g8_of(mode(o8), da, mha.out_proj.weight, bias=mha.out_proj.bias)
        do = lin_dx(da, mha.out_proj.weight, fp8=mode(o8), g8=g8).view(b, Lq, d)
        lin_dw(da, _rows(o), mha.out_proj.weight, mha.out_proj.bias, g8=g8, x8=o8)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        K.attn_bwd(q, kv[..., :d], kv[..., d:], o, do, lse, layer.heads, kmask, dq=dq, dk=dkv[..., :d],
                   dv=dkv[..., d:], drop_p=pd, seed=seeds[0])
        g8 = g8_of(mode(tq8), _rows(dq), wq, bias=bq)
        dtgt = lin_dx(_rows(dq), wq, fp8=mode(tq8), g8=g8, residual=gs).view(b, Lq, d) if ctx.needs_input_grad[0] else None
        lin_dw(_rows(dq), _rows(tgt), wq, bq, g8=g8, x8=tq8)
        g8 = g8_of(mode(mem8), _rows(dkv), wkv, bias=bkv)
        dmem = lin_dx(_rows(dkv), wkv, fp8=mode(mem8), g8=g8).view(b, Lk, d) if ctx.needs_input_grad[1] else None
        lin_dw(_rows(dkv), _rows(memory), wkv, bkv, g8=g8, x8=mem8)
        return (dtgt, dmem, None, None) + (None,) * len(ctx.needs_input_grad[4:])


# --------------------------------------------------------------------------------------- losses
# The loss kernels take their upstream gradient as a host float.  Reading it from the incoming 0-dim tensor (float(g)) would drain
# the stream - once per loss term, eight times at the very start of every backward pass, with the GPU idling until the host has
# enqueued work again - so the kernels run with 1.0 and the (tiny, [G, D]) results are scaled by g on the device.
def _scaled(g, *ts):
    return tuple(None if t is None else t * g.to(t.dtype) for t in ts)


class SimInfoNCEFn(torch.autograd.Function):
    """get_contrastive_loss's compute (models/xvlm.py:273-292) on already gathered f32 features."""

    @staticmethod
    def forward(ctx, a, b, temp, idx, both_dirs, tval=None, local_rows=None):
        """local_rows = (row0, nrows): a and b are all-gathered and only those rows are this rank's (AllGather keeps the
        local slice of the gradient, xvlm.py:62-67): the backward computes just them - 1/W of its work."""
        a, b = a.contiguous(), b.contiguous()
        if tval is None:       # the learnable temperature is read by the kernels ON THE DEVICE (no float(temp): that drains the stream)
            tval = temp.data.view(1) if temp is not None else 1.0
        loss, ws = K.sim_infonce_fwd(a, b, idx, tval, both_dirs)
        ctx.save_for_backward(a, b, ws)
        ctx.meta = (tval, idx, both_dirs, temp is not None, local_rows)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        a, b, ws = ctx.saved_tensors
        tval, idx, both_dirs, has_temp, rows = ctx.meta
        dtemp = torch.zeros(1, device=a.device, dtype=torch.float32) if has_temp else None
        da, db = K.sim_infonce_bwd(a, b, idx, tval, ws, 1.0, both_dirs, dtemp, rows)
        if has_temp and rows is not None and rows[1] != a.shape[0]:
            # temp sits after the gather: every rank must hold the FULL dL/dtemp (DDP then averages identical values).
            # The row shards' shares add up to it over the ranks, so W x share survives the mean.
            dtemp = dtemp * (a.shape[0] / rows[1])
        da, db, dtemp = _scaled(g, da, db, dtemp)
        return da, db, (dtemp[0] if has_temp else None), None, None, None, None


class CapSimFn(torch.autograd.Function):
    """get_caption_contrastive_loss (model_retrieval_caption.py:145-152); c [b,n,D] batch-major, t [b,D]."""

    @staticmethod
    def forward(ctx, c, t, temp, tval=None):
        c, t = c.contiguous(), t.contiguous()
        tval = temp.data.view(1) if tval is None else tval          # device scalar, read by the kernels
        loss, am, ws = K.capsim_fwd(c, t, tval)
        ctx.save_for_backward(c, t, am, ws)
        ctx.tval = tval
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        c, t, am, ws = ctx.saved_tensors
        dtemp = torch.zeros(1, device=c.device, dtype=torch.float32)
        dc, dt = K.capsim_bwd(c, t, ctx.tval, am, ws, 1.0, dtemp)
        dc, dt, dtemp = _scaled(g, dc, dt, dtemp)
        return dc, dt, dtemp[0], None


class DstlFn(torch.autograd.Function):
    """dstl_loss's compute (model_retrieval_caption.py:100-116) on gathered f32 features; grads to v and t only."""

    @staticmethod
    def forward(ctx, v, c, s, t, alpha, local_rows=None):
        v, c, s, t = v.contiguous(), c.contiguous(), s.contiguous(), t.contiguous()
        loss, ws = K.dstl_fwd(v, c, s, t, alpha)
        ctx.save_for_backward(v, t, ws)
        ctx.rows = local_rows
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        v, t, ws = ctx.saved_tensors
        dv, dt = _scaled(g, *K.dstl_bwd(v, t, ws, 1.0, ctx.rows))
        return dv, None, None, dt, None, None


class CvPoolFn(torch.autograd.Function):
    """mean over tokens of F.normalize(x, dim=1) (the token axis!) - caption_vision_loss in pooled form."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        pooled, inv = K.cvpool_fwd(x)
        ctx.save_for_backward(x, inv)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        x, inv = ctx.saved_tensors
        return K.cvpool_bwd(x, inv, dp.contiguous())


class CapRegFn(torch.autograd.Function):
    """caption_regularization (model_retrieval_caption.py:65-73); c [b,n,d] batch-major."""

    @staticmethod
    def forward(ctx, c):
        c = c.contiguous()
        ctx.save_for_backward(c)
        return K.capreg_fwd(c)[0]

    @staticmethod
    def backward(ctx, g):
        (c,) = ctx.saved_tensors
        return _scaled(g, K.capreg_bwd(c, 1.0))[0]
