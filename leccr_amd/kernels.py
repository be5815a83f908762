"""Thin functional wrappers over the C ABI (one Python function per entry point, tensors in / tensors out).

No autograd here (see ops.py) and no fallback: every function launches a HIP kernel from libleccr_hip.so on the
current torch stream.  torch is used for allocation only.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from ._lib import (BF16, EPI_DGELU, EPI_DQUICKGELU, EPI_GELU, EPI_NONE, EPI_QUICKGELU, F32, FP8, FP8_BF8A, check, dtype_code, lib,
                   ptr, stream)


def _rows2d(t):
    """View [.., d] as (rows, d, ld) with a unit inner stride."""
    assert t.stride(-1) == 1, 'inner dimension must be contiguous'
    t2 = t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t
    assert t2.dim() == 2
    return t2, t2.shape[0], t2.shape[1], t2.stride(0)


# ---- fp32 operands on the bf16 matrix cores (leccr_split3: x = hi + lo, three bf16 products in one launch over 3 K) ----------------
# The reference's arithmetic is fp32 (models/xvlm.py:88).  With f32_mfma (default; LECCR_F32_MFMA=0 or set_f32_mfma(False): the scalar
# f32 kernels, correctness yardstick) every f32 gemm() of a shape the MFMA kernels take runs as ONE bf16 product over hi / lo splits of
# its operands, f32 accumulation and output: ~2^-17 relative per product term instead of bf16's 2^-9.
FORM_X, FORM_G = 0, 1          # hi|hi|lo (activations, [in, out] weight copies)   /   hi|lo|hi (gradients, [out, in] weights)
f32_mfma = os.environ.get('LECCR_F32_MFMA', '1') != '0'
f32_split_launches = 0         # gemm() calls that took the split path (tests read it)
# segments per copy: 3 (hi | hi | lo against hi | lo | hi: 2^-17 per product term, the default) or 6 (three bf16 pieces, six products:
# 2^-24, fp32-grade, twice the matrix-core work; LECCR_F32_SPLITS=6 / set_f32_segments(6))
NS = 6 if os.environ.get('LECCR_F32_SPLITS', '3') == '6' else 3


def set_f32_segments(n):
    global NS
    assert n in (3, 6)
    prev, NS = NS, n
    return prev
_SPLIT_MIN_WORK = 1 << 27      # M * N * K below which the exact scalar kernel is as fast (the 256-wide heads in front of the 1 / temp
                               # amplified losses among them: [b, 768] x [256, 768] at b = 256 is 2^25.6)


def set_f32_mfma(on):
    global f32_mfma
    prev, f32_mfma = f32_mfma, bool(on)
    return prev


def split3(x, form, transpose=False, colsum=None):
    """f32 [rows, cols] (unit inner stride) -> bf16 [rows, 3 * cols] (transpose: [cols, 3 * rows]) in hi / lo form `form`.
    colsum ([cols] f32): += the column sums of x, in the same pass (the bias gradient beside a dY's split)."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and form in (FORM_X, FORM_G)
    rows, cols = x.shape
    out = torch.empty((cols, NS * rows) if transpose else (rows, NS * cols), device=x.device, dtype=torch.bfloat16)
    if colsum is not None:
        assert colsum.dtype == torch.float32 and colsum.numel() == cols and colsum.is_contiguous() and not transpose
    check(lib().leccr_split_n(ptr(x), ptr(out), rows, cols, x.stride(0), form, NS, int(transpose), ptr(colsum), stream()), 'leccr_split_n')
    return out


def gemm256_takes(M, N, Kd):
    """Shapes of an NT product the 8-wave 256 x 256 kernel takes (gemm256_wanted in gemm256.hip): the ones whose epilogue can write
    the split copy c3 without a C."""
    return M >= 256 and N >= 256 and N % 8 == 0 and Kd >= 128 and Kd % 64 == 0 and -(-M // 256) * -(-N // 256) >= 192


def tn4_takes(M, N, Kd):
    """Shapes of a bf16 weight-gradient product (C [M, N] f32 += A[Kd, M]^T B[Kd, N]) the 4-wave TN kernel takes (gemm256t4_legal in gemm256t4.hip,
    contiguous operands): the launches whose column sums (asum / asum_x3) ride inside the product."""
    return (M >= 256 and N >= 256 and M % 256 == 0 and N % 256 == 0 and Kd % 64 == 0 and Kd >= 4096 and max(M, N) < (1 << 22)
            and os.environ.get('LECCR_GEMM_TN4', '1')[:1] != '0')


def split3_colsum_ok(x):
    """Shapes whose split pass can carry the column sums (leccr_split3's colsum)."""
    return NS == 6 or (x.shape[1] % 8 == 0 and x.shape[1] <= 8192 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def _split_plan(a, b, trans_a, trans_b, out, accumulate, asum):
    """Which split layout an f32 product takes on the MFMA kernels, or None (the scalar kernel): 'nt' / 'ntb' (B stored [K, N]:
    transposed split) / 'tn' (reduction over rows: the [rows, 3 cols] copies read as [3 rows, cols])."""
    if not (f32_mfma and a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_cuda):
        return None
    if out is not None and out.dtype != torch.float32:
        return None
    M, Kd = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N = b.shape[1] if trans_b else b.shape[0]
    if M * N * Kd < _SPLIT_MIN_WORK or N % 4:
        return None
    if not trans_a:
        return ('ntb' if trans_b else 'nt') if Kd % 64 == 0 else None
    if trans_b and M % 8 == 0 and N % 8 == 0 and a.stride(0) == M and b.stride(0) == N:
        return 'tn'
    return None


def gemm(a, b, *, trans_a=False, trans_b=False, bias=None, epilogue=EPI_NONE, aux=None, aux_out=None, residual=None,
         out=None, out_dtype=None, accumulate=False, alpha=1.0, colsum=None, asum=None, alpha_dev=None, c8=None, write_c=True,
         a3=None, b3=None, c3_form=None, asum_x3=False):
    """C = epilogue(alpha * A' B'^T).  a: [M,K] (or [K,M] if trans_a); b: [N,K] (or [K,N] if trans_b); 2-D, row-major.
    write_c=False (fp8 operands with c8 only): the bf16 C is not written - only its fp8 copy c8 (and aux_out / colsum); returns None.
    f32 operands: see f32_mfma above.  a3 / b3 = (split copy, form): the caller already holds leccr_split3 copies of a / b (as they
    are stored; for trans_b alone b3 is the transposed split) - they must be of opposite forms; missing ones are made here.
    c3_form (bf16 operands, f32 C, NT): also return the bf16 (hi, lo) split copy [M, 3 N] of C in that form, written by the kernel's
    epilogue where it can (else by a leccr_split3 pass inside leccr_gemm): -> (C, C3); write_c=False then leaves C itself unwritten
    (None) - only for shapes gemm256_takes()."""
    plan = _split_plan(a, b, trans_a, trans_b, out, accumulate, asum) if a.dtype == torch.float32 else None
    assert c3_form is None or a.dtype == torch.bfloat16 or plan in ('nt', 'ntb'), 'c3 exists for the bf16 / split-operand NT products only'
    if plan is not None:
        global f32_split_launches
        f32_split_launches += 1
        fa = a3[1] if a3 is not None else (1 - b3[1] if b3 is not None else FORM_X)
        assert b3 is None or b3[1] == 1 - fa
        a_s = a3[0] if a3 is not None else split3(a, fa)
        b_s = b3[0] if b3 is not None else split3(b, 1 - fa, transpose=(plan == 'ntb'))
        if plan == 'tn':
            a_s, b_s = a_s.view(NS * a.shape[0], a.shape[1]), b_s.view(NS * b.shape[0], b.shape[1])
            if asum is not None:        # the column sums of the split copy would count hi twice: one pass over the f32 operand instead
                if alpha == 1.0:
                    colsum_(a, asum)
                else:
                    tmp = torch.zeros_like(asum)
                    colsum_(a, tmp)
                    axpy(tmp, asum, alpha)
            return gemm(a_s, b_s, trans_a=True, trans_b=True, out=out, out_dtype=torch.float32, accumulate=accumulate, alpha=alpha,
                        colsum=colsum)
        assert a_s.shape == (a.shape[0], NS * a.shape[1]) and b_s.shape[1] == a_s.shape[1]
        return gemm(a_s, b_s, bias=bias, epilogue=epilogue, aux=aux, aux_out=aux_out, residual=residual, out=out,
                    out_dtype=torch.float32, accumulate=accumulate, alpha=alpha, colsum=colsum, c3_form=c3_form, write_c=write_c)
    assert a.dim() == 2 and b.dim() == 2 and (a.dtype == b.dtype or a.dtype == torch.float8_e5m2)
    assert a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    N, Kb = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    fp8 = a.dtype in (torch.float8_e4m3fn, torch.float8_e5m2)
    if not write_c:
        assert out is None and not trans_a and ((fp8 and c8 is not None) or (c3_form is not None and gemm256_takes(M, N, K)))
    elif out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype or (torch.bfloat16 if fp8 else a.dtype))
    assert out is None or (out.shape == (M, N) and out.stride(1) == 1)
    p = L.GemmParams()
    p.A, p.B, p.C = ptr(a), ptr(b), ptr(out)
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc = a.stride(0), b.stride(0), (out.stride(0) if out is not None else N)
    if fp8:     # fp8 operands (quantize_fp8): b e4m3, a e4m3 (forward) or e5m2 (gradients); alpha_dev = their two inverse scales
        assert b.dtype == torch.float8_e4m3fn and trans_a == trans_b      # NT (forward / dX) or the weight-gradient form dY^T X
        p.ab_dtype, p.c_dtype = (FP8 if a.dtype == torch.float8_e4m3fn else FP8_BF8A), (dtype_code(out) if out is not None else BF16)
    elif out is None:       # c3 without a C: bf16 operands, the (unwritten) C counts as f32
        p.ab_dtype, p.c_dtype = dtype_code(a), F32
    else:
        p.ab_dtype, p.c_dtype = dtype_code(a), dtype_code(out)
    if alpha_dev is not None:
        for i, t in enumerate(alpha_dev):
            assert t.dtype == torch.float32 and t.numel() >= 1 and t.is_cuda
            p.alpha_dev[i] = t.data_ptr()
    p.trans_a, p.trans_b = int(trans_a), int(trans_b)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    p.bias = ptr(bias)
    p.epilogue, p.accumulate = epilogue, int(accumulate)
    odt = out.dtype if out is not None else (torch.float32 if c3_form is not None else torch.bfloat16)
    for t in (aux, aux_out, residual):
        if t is not None:
            assert t.dtype == odt and t.shape == (M, N) and t.stride(1) == 1
    p.aux, p.ldaux = ptr(aux), (aux.stride(0) if aux is not None else 0)
    if aux_out is not None:
        assert aux_out.stride(0) == p.ldc
    p.aux_out = ptr(aux_out)
    p.residual, p.ldr = ptr(residual), (residual.stride(0) if residual is not None else 0)
    p.alpha = alpha
    if colsum is not None:
        assert colsum.dtype == torch.float32 and colsum.numel() == N and colsum.is_contiguous() and not accumulate
    p.colsum = ptr(colsum)
    if asum is not None:        # [M] f32 += alpha * column sums of the stored [K, M] A over K (the bias gradient beside a dW product)
        assert trans_a and asum.dtype == torch.float32 and asum.numel() == M and asum.is_contiguous()
    p.asum = ptr(asum)
    if asum_x3:                 # a = the G-form split copy of an f32 dY read as [3 rows, M]: asum takes its hi and lo segments once each (leccr_hip.h)
        assert asum is not None and a.dtype == torch.bfloat16 and trans_a and K % 3 == 0 and NS == 3
        p.asum_x3 = 1
    if c8 is not None:          # (fp8 tensor [M, N], Fp8Site): the fused fp8 copy of C for the product that consumes it
        t8, site8 = c8
        assert t8.shape == (M, N) and t8.stride(1) == 1 and t8.element_size() == 1
        p.c8, p.ldc8, p.c8_site, p.c8_fmt = ptr(t8), t8.stride(0), ptr(site8.row), site8.fmt
    c3 = None
    if c3_form is not None:
        assert not fp8 and a.dtype == torch.bfloat16 and (out is None or out.dtype == torch.float32)
        assert not trans_a and not trans_b and not accumulate and c3_form in (FORM_X, FORM_G) and N % 4 == 0 and NS == 3
        c3 = torch.empty(M, 3 * N, device=a.device, dtype=torch.bfloat16)
        p.c3, p.c3_form = ptr(c3), c3_form
    check(lib().leccr_gemm(C.byref(p), stream()), 'leccr_gemm')
    return (out, c3) if c3_form is not None else out


def _q8_out(site, rows, d, device):
    """(fp8 buffer, site row pointer, fmt) for a fused fp8 copy under `site`, or (None, None, 0) while the site has no scale yet."""
    if site is None or not site.primed or d % 8:
        return None, None, 0
    return torch.empty(rows, d, device=device, dtype=site.torch_dtype), site.row, site.fmt


def layernorm_fwd(x, gamma, beta, eps, add=None, want_sum=False, drop=None, q8_site=None, s3_form=None, write_y=True):
    """drop = (p, seed): normalise dropout(x) (+ add) - the mask leccr_dropout(x, p, seed) would apply, inside this pass.
    q8_site (an Fp8Site): also return y's fp8 copy under that site as a 5th value (None until the site is primed); write_y=False (primed site):
    only that copy is written, y comes back None.
    s3_form (f32 rows, 3-segment copies): also return y's bf16 (hi, lo) split copy [rows, 3 d] in that form as a 5th value, written by the
    pass itself; write_y=False then leaves y itself unwritten (None) - for a tensor nothing else reads."""
    x2, rows, d, ld = _rows2d(x)
    assert ld == d
    if s3_form is not None:
        assert x.dtype == torch.float32 and q8_site is None and NS == 3 and s3_form in (FORM_X, FORM_G)
        y = torch.empty_like(x2) if write_y else None
        y3 = torch.empty(rows, 3 * d, device=x.device, dtype=torch.bfloat16)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        s = torch.empty_like(x2) if want_sum else None
        if add is not None:
            add = add.reshape(rows, d)
            assert add.is_contiguous() and add.dtype == x.dtype
        p_, seed_ = (float(drop[0]), int(drop[1])) if drop is not None else (0.0, 0)
        check(lib().leccr_layernorm_fwd_s3(ptr(x2), ptr(add), ptr(gamma), ptr(beta), ptr(y), ptr(s), ptr(mean), ptr(rstd), rows, d, eps,
                                           p_, seed_, ptr(y3), s3_form, stream()), 'leccr_layernorm_fwd_s3')
        return (y.view(x.shape) if y is not None else None), mean, rstd, (s.view(x.shape) if s is not None else None), y3
    y = torch.empty_like(x2) if write_y else None          # (write_y=False: beside a fused fp8 copy only - checked below)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    s = torch.empty_like(x2) if want_sum else None
    if add is not None:
        add = add.reshape(rows, d)
        assert add.is_contiguous() and add.dtype == x.dtype
    if q8_site is not None:
        y8, site_row, fmt = _q8_out(q8_site, rows, d, x.device)
        assert write_y or y8 is not None, 'write_y=False needs a primed site (the fp8 copy is then the only output)'
        p_, seed_ = (float(drop[0]), int(drop[1])) if drop is not None else (0.0, 0)
        check(lib().leccr_layernorm_fwd_ex(ptr(x2), ptr(add), ptr(gamma), ptr(beta), ptr(y), ptr(s), ptr(mean), ptr(rstd),
                                           rows, d, eps, p_, seed_, ptr(y8), ptr(site_row), fmt, dtype_code(x), stream()),
              'leccr_layernorm_fwd_ex')
        return (y.view(x.shape) if y is not None else None), mean, rstd, (s.view(x.shape) if s is not None else None), y8
    assert write_y
    if drop is not None and drop[0] > 0:
        check(lib().leccr_layernorm_fwd_drop(ptr(x2), ptr(add), ptr(gamma), ptr(beta), ptr(y), ptr(s), ptr(mean), ptr(rstd),
                                             rows, d, eps, float(drop[0]), int(drop[1]), dtype_code(x), stream()),
              'leccr_layernorm_fwd_drop')
    else:
        check(lib().leccr_layernorm_fwd(ptr(x2), ptr(add), ptr(gamma), ptr(beta), ptr(y), ptr(s), ptr(mean), ptr(rstd),
                                        rows, d, eps, dtype_code(x), stream()), 'leccr_layernorm_fwd')
    return y.view(x.shape), mean, rstd, (s.view(x.shape) if s is not None else None)


def layernorm_bwd(dy, x_in, gamma, mean, rstd, dgamma, dbeta, dres=None, colsum=None, colsum_of=0, drop=None, q8_site=None, s3_form=None):
    """colsum_of: 1 = colsum += column sums of dres, 2 = of the returned dx (fused bias gradient, see leccr_hip.h).
    drop = (p, seed): returns (dx, dropout(dx; p, seed)) from one pass.
    q8_site (an Fp8Site): returns (dx, dropout(dx) or None, g8): g8 = the fp8 copy of dropout(dx) (of dx without drop) under that
    site, None until the site is primed.
    s3_form (f32 rows, 3-segment copies): returns (dx, dropout(dx) or None, g3): g3 [rows, 3 d] = the bf16 (hi, lo) split copy of dropout(dx) (of dx
    without drop) in that form, written by the pass itself."""
    dy2, rows, d, ld = _rows2d(dy)
    assert ld == d and x_in.is_contiguous()
    dx = torch.empty_like(dy2)
    if dres is not None:
        assert dres.is_contiguous() and dres.dtype == dy.dtype
    if s3_form is not None:
        assert q8_site is None and not colsum_of and dy.dtype == torch.float32 and NS == 3 and s3_form in (FORM_X, FORM_G) and d % 4 == 0
        g3 = torch.empty(rows, 3 * d, device=dy.device, dtype=torch.bfloat16)
        dxd = torch.empty_like(dy2) if drop is not None else None
        p_, seed_ = (float(drop[0]), int(drop[1])) if drop is not None else (0.0, 0)
        check(lib().leccr_layernorm_bwd_ex(ptr(dy2), ptr(x_in), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                                           ptr(dgamma), ptr(dbeta), ptr(dxd), p_, seed_, ptr(g3), None, 2 + s3_form, rows, d,
                                           dtype_code(dy), stream()), 'leccr_layernorm_bwd_ex')
        return dx.view(dy.shape), (dxd.view(dy.shape) if dxd is not None else None), g3
    if q8_site is not None:
        assert not colsum_of
        g8, site_row, fmt = _q8_out(q8_site, rows, d, dy.device)
        dxd = torch.empty_like(dy2) if drop is not None else None
        p_, seed_ = (float(drop[0]), int(drop[1])) if drop is not None else (0.0, 0)
        check(lib().leccr_layernorm_bwd_ex(ptr(dy2), ptr(x_in), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                                           ptr(dgamma), ptr(dbeta), ptr(dxd), p_, seed_, ptr(g8), ptr(site_row), fmt, rows, d,
                                           dtype_code(dy), stream()), 'leccr_layernorm_bwd_ex')
        return dx.view(dy.shape), (dxd.view(dy.shape) if dxd is not None else None), g8
    if drop is not None:
        assert not colsum_of
        dxd = torch.empty_like(dy2)
        check(lib().leccr_layernorm_bwd_drop(ptr(dy2), ptr(x_in), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                                             ptr(dgamma), ptr(dbeta), ptr(dxd), float(drop[0]), int(drop[1]), rows, d,
                                             dtype_code(dy), stream()), 'leccr_layernorm_bwd_drop')
        return dx.view(dy.shape), dxd.view(dy.shape)
    if colsum_of:
        assert colsum is not None and colsum.dtype == torch.float32 and colsum.numel() == d and colsum.is_contiguous()
    check(lib().leccr_layernorm_bwd(ptr(dy2), ptr(x_in), ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), ptr(dx),
                                    ptr(dgamma), ptr(dbeta), ptr(colsum), int(colsum_of), rows, d, dtype_code(dy),
                                    stream()), 'leccr_layernorm_bwd')
    return dx.view(dy.shape)


def _attn_args(q, k, v):
    # q [b, Lq, heads*hd] possibly a strided view of a packed projection (row stride = ld, batch stride = Lq*ld)
    for t, Lx in ((q, q.shape[1]), (k, k.shape[1]), (v, v.shape[1])):
        assert t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == Lx * t.stride(1), (t.shape, t.stride())
    return q.shape[0], q.shape[1], k.shape[1]


def attn_fwd(q, k, v, heads, kmask=None, drop_p=0.0, seed=0, q8_site=None, causal=False):
    """q8_site (a primed Fp8Site, bf16 only): returns (o, lse, o8) with o8 [b * Lq, d] the e4m3 copy of o under that site.
    causal: query i sees keys j <= i only (self-attention, no dropout: CLIP's text transformer)."""
    b, Lq, Lk = _attn_args(q, k, v)
    d = q.shape[2]
    hd = d // heads
    o = torch.empty(b, Lq, d, device=q.device, dtype=q.dtype)
    lse = torch.empty(b, heads, Lq, device=q.device, dtype=torch.float32)
    if kmask is not None:
        assert kmask.dtype == torch.uint8 and kmask.shape == (b, Lk) and kmask.is_contiguous()
    if causal:
        assert Lq == Lk and drop_p == 0.0 and q8_site is None
        check(lib().leccr_attn_fwd_causal(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(kmask), b, heads, Lq, hd,
                                          q.stride(1), k.stride(1), v.stride(1), o.stride(1), dtype_code(q), stream()), 'leccr_attn_fwd_causal')
        return o, lse
    if q8_site is not None:
        assert q8_site.primed and q8_site.fmt == 0 and d % 8 == 0
        o8 = torch.empty(b * Lq, d, device=q.device, dtype=q8_site.torch_dtype)
        check(lib().leccr_attn_fwd_q8(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(kmask), b, heads, Lq, Lk, hd,
                                      q.stride(1), k.stride(1), v.stride(1), o.stride(1), dtype_code(q), float(drop_p),
                                      int(seed), ptr(o8), d, ptr(q8_site.row), stream()), 'leccr_attn_fwd_q8')
        return o, lse, o8
    check(lib().leccr_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(lse), ptr(kmask), b, heads, Lq, Lk, hd,
                               q.stride(1), k.stride(1), v.stride(1), o.stride(1), dtype_code(q), float(drop_p),
                               int(seed), stream()), 'leccr_attn_fwd')
    return o, lse


def attn_s3_takes(q, k, v, heads):
    """Whether attn_fwd_s3 / attn_bwd_s3 take the problem: f32 self-attention (packed qkv views, Lq == Lk) on the pair-form kernels - head size
    64, no causal mask, copy rows 16-byte aligned (leccr_attn_s3_takes)."""
    if q.dtype != torch.float32 or not q.is_cuda or NS != 3:
        return False
    b, Lq, Lk = _attn_args(q, k, v)
    d = q.shape[2]
    return Lq == Lk and d % heads == 0 and bool(lib().leccr_attn_s3_takes(b, heads, Lq, Lk, d // heads, q.stride(1), k.stride(1), v.stride(1), 9 * d, 3 * d)) \
        and bool(lib().leccr_attn_s3_takes(b, heads, Lq, Lk, d // heads, q.stride(1), k.stride(1), v.stride(1), 3 * d, d))


def attn_fwd_s3(q, k, v, heads, kmask=None, drop_p=0.0, seed=0):
    """f32 attention whose output exists only as its X-form operand copy (leccr_split3's layout): -> (o3 [b * Lq, 3 d] bf16, lse)."""
    b, Lq, Lk = _attn_args(q, k, v)
    d = q.shape[2]
    o3 = torch.empty(b * Lq, 3 * d, device=q.device, dtype=torch.bfloat16)
    lse = torch.empty(b, heads, Lq, device=q.device, dtype=torch.float32)
    if kmask is not None:
        assert kmask.dtype == torch.uint8 and kmask.shape == (b, Lk) and kmask.is_contiguous()
    check(lib().leccr_attn_fwd_s3(ptr(q), ptr(k), ptr(v), ptr(o3), ptr(lse), ptr(kmask), b, heads, Lq, Lk, d // heads,
                                  q.stride(1), k.stride(1), v.stride(1), 3 * d, d, float(drop_p), int(seed), stream()), 'leccr_attn_fwd_s3')
    return o3, lse


def attn_bwd_s3(q, k, v, o3, dout, lse, heads, kmask=None, drop_p=0.0, seed=0, bias_sum=None):
    """Backward of attn_fwd_s3 for self-attention on a packed qkv: reads o as its copy o3, writes the gradient of the packed [rows, 3 d] tensor
    only as its G-form operand copy -> dqkv3 [b * L, 3 * 3 d] bf16 (segments hi | lo | hi, each [dq | dk | dv]).  bias_sum ([3 d] f32): += the
    column sums of that gradient (the packed in-projection's bias gradient), summed inside the kernels."""
    b, Lq, Lk = _attn_args(q, k, v)
    d = q.shape[2]
    assert Lq == Lk and dout.is_contiguous() and dout.shape == (b, Lq, d) and o3.shape == (b * Lq, 3 * d) and o3.is_contiguous()
    assert bias_sum is None or (bias_sum.dtype == torch.float32 and bias_sum.numel() == 3 * d and bias_sum.is_contiguous())
    dqkv3 = torch.empty(b * Lq, 9 * d, device=q.device, dtype=torch.bfloat16)
    delta = torch.empty(2 * b * heads * ((Lq + 63) // 64 * 64), dtype=torch.float32, device=q.device)
    if kmask is not None:
        assert kmask.dtype == torch.uint8 and kmask.shape == (b, Lk) and kmask.is_contiguous()
    check(lib().leccr_attn_bwd_s3(ptr(q), ptr(k), ptr(v), ptr(o3), 3 * d, d, ptr(dout), ptr(lse), ptr(kmask),
                                  ptr(dqkv3), ptr(dqkv3[:, d:2 * d]), ptr(dqkv3[:, 2 * d:3 * d]), ptr(delta), b, heads, Lq, Lk, d // heads,
                                  q.stride(1), k.stride(1), v.stride(1), dout.stride(1), 9 * d, 3 * d, ptr(bias_sum), float(drop_p), int(seed), stream()),
          'leccr_attn_bwd_s3')
    return dqkv3


def attn_bwd(q, k, v, o, dout, lse, heads, kmask=None, dq=None, dk=None, dv=None, drop_p=0.0, seed=0, q8=None, causal=False, write_bf16=True):
    """q8 = (dq8, dk8, dv8, site): e5m2 copies of the three gradients (views of one packed [b, L, 3 d] byte tensor) under one primed site.
    write_bf16=False (with q8, head size 64, bf16): only those copies are written -> (None, None, None)."""
    b, Lq, Lk = _attn_args(q, k, v)
    d = q.shape[2]
    hd = d // heads
    assert o.is_contiguous() and dout.is_contiguous() and dout.shape == o.shape
    # gradients must have the strides of their inputs (the model passes views of one packed dqkv buffer)
    if write_bf16:
        dq = torch.empty_strided(q.shape, q.stride(), device=q.device, dtype=q.dtype) if dq is None else dq
        dk = torch.empty_strided(k.shape, k.stride(), device=q.device, dtype=q.dtype) if dk is None else dk
        dv = torch.empty_strided(v.shape, v.stride(), device=q.device, dtype=q.dtype) if dv is None else dv
        assert dq.stride(1) == q.stride(1) and dk.stride(1) == k.stride(1) and dv.stride(1) == v.stride(1)
    else:
        assert q8 is not None and not causal and dq is None and dk is None and dv is None
    # workspace: two rows of constants per (b, head, query), queries padded to a multiple of 64 (see leccr_hip.h)
    delta = torch.empty(2 * b * heads * ((Lq + 63) // 64 * 64), dtype=torch.float32, device=q.device)
    if causal:
        assert Lq == Lk and drop_p == 0.0 and q8 is None
        check(lib().leccr_attn_bwd_causal(ptr(q), ptr(k), ptr(v), ptr(o), ptr(dout), ptr(lse), ptr(kmask), ptr(dq), ptr(dk), ptr(dv),
                                          ptr(delta), b, heads, Lq, hd, q.stride(1), k.stride(1), v.stride(1), o.stride(1),
                                          dtype_code(q), stream()), 'leccr_attn_bwd_causal')
        return dq, dk, dv
    if q8 is not None:
        dq8, dk8, dv8, site = q8
        assert site.primed and site.fmt == 1 and dq8.stride(1) == dk8.stride(1) == dv8.stride(1) and dq8.stride(2) == 1
        assert dq8.shape == q.shape and dk8.shape == k.shape and dv8.shape == v.shape and dq8.element_size() == 1
        check(lib().leccr_attn_bwd_q8(ptr(q), ptr(k), ptr(v), ptr(o), ptr(dout), ptr(lse), ptr(kmask), ptr(dq), ptr(dk),
                                      ptr(dv), ptr(delta), b, heads, Lq, Lk, hd, q.stride(1), k.stride(1), v.stride(1),
                                      o.stride(1), dtype_code(q), float(drop_p), int(seed), ptr(dq8), ptr(dk8), ptr(dv8),
                                      dq8.stride(1), ptr(site.row), stream()), 'leccr_attn_bwd_q8')
        return dq, dk, dv
    check(lib().leccr_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(o), ptr(dout), ptr(lse), ptr(kmask), ptr(dq), ptr(dk),
                               ptr(dv), ptr(delta), b, heads, Lq, Lk, hd, q.stride(1), k.stride(1), v.stride(1),
                               o.stride(1), dtype_code(q), float(drop_p), int(seed), stream()), 'leccr_attn_bwd')
    return dq, dk, dv


def colsum_(x, out):
    return colsum(x, out)


def colsum(x, out):
    """out[n] += sum_m x[m, n]  (bias gradients)."""
    x2, M, N, ld = _rows2d(x)
    assert out.dtype == torch.float32 and out.numel() == N
    check(lib().leccr_colsum(ptr(x2), ptr(out), M, N, ld, dtype_code(x), stream()), 'leccr_colsum')
    return out


def l2norm_fwd(x):
    x2, rows, d, ld = _rows2d(x)
    assert ld == d
    y = torch.empty_like(x2)
    inv = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib().leccr_l2norm_fwd(ptr(x2), ptr(y), ptr(inv), rows, d, dtype_code(x), stream()), 'leccr_l2norm_fwd')
    return y.view(x.shape), inv


def l2norm_bwd(dy, y, inv):
    dy2, rows, d, ld = _rows2d(dy)
    assert ld == d and y.is_contiguous()
    dx = torch.empty_like(dy2)
    check(lib().leccr_l2norm_bwd(ptr(dy2), ptr(y), ptr(inv), ptr(dx), rows, d, dtype_code(dy), stream()),
          'leccr_l2norm_bwd')
    return dx.view(dy.shape)


# --------------------------------------------------------------------------------------- tower input stages
def patchify(image, p, Kp, dtype):
    b, c, R, _ = image.shape
    assert c == 3 and image.dtype == torch.float32 and image.is_contiguous()
    g = R // p
    out = torch.empty(b * g * g, Kp, device=image.device, dtype=dtype)
    check(lib().leccr_patchify(ptr(image), ptr(out), b, R, p, Kp, dtype_code(out), stream()), 'leccr_patchify')
    return out


def vit_assemble_fwd(patch_out, cls, pos, b):
    N, d = pos.shape
    tok = torch.empty(b, N, d, device=patch_out.device, dtype=patch_out.dtype)
    check(lib().leccr_vit_assemble_fwd(ptr(patch_out), ptr(cls), ptr(pos), ptr(tok), b, N, d, dtype_code(tok),
                                       stream()), 'leccr_vit_assemble_fwd')
    return tok


def vit_assemble_bwd(dtok, dcls, dpos):
    b, N, d = dtok.shape
    assert dtok.is_contiguous()
    dpatch = torch.empty(b * (N - 1), d, device=dtok.device, dtype=dtok.dtype)
    check(lib().leccr_vit_assemble_bwd(ptr(dtok), ptr(dpatch), ptr(dcls), ptr(dpos), b, N, d, dtype_code(dtok),
                                       stream()), 'leccr_vit_assemble_bwd')
    return dpatch


def bert_embed_fwd(ids, word, pos, type0, dtype):
    b, Lx = ids.shape
    vocab, d = word.shape
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    emb = torch.empty(b, Lx, d, device=ids.device, dtype=dtype)
    check(lib().leccr_bert_embed_fwd(ptr(ids), ptr(word), ptr(pos), ptr(type0), ptr(emb), b, Lx, d, vocab,
                                     dtype_code(emb), stream()), 'leccr_bert_embed_fwd')
    return emb


def bert_embed_bwd(ids, demb, dword, dpos, dtype0, pad_id=0):
    b, Lx, d = demb.shape
    assert demb.is_contiguous()
    check(lib().leccr_bert_embed_bwd(ptr(ids), ptr(demb), ptr(dword), ptr(dpos), ptr(dtype0), b, Lx, d,
                                     dword.shape[0], pad_id, dtype_code(demb), stream()), 'leccr_bert_embed_bwd')


# --------------------------------------------------------------------------------------- losses (f32)
def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.stride())
    return t


def _temp_args(temp):
    """(host value, device pointer) of a temperature argument: a float, or a 1-element f32 DEVICE tensor the kernels read at run time
    (the learnable XVLMBase.temp: no host read of a device scalar in the training step)."""
    if torch.is_tensor(temp):
        assert temp.is_cuda and temp.dtype == torch.float32 and temp.numel() == 1, (temp.device, temp.dtype, temp.shape)
        return 1.0, ptr(temp)
    return float(temp), None


def sim_infonce_fwd(a, b, idx, temp, both_dirs=True):
    G, D = _f32c(a).shape
    _f32c(b)
    loss = torch.empty(1, device=a.device, dtype=torch.float32)
    # row / column log-sum-exps (what the backward reads) | per-tile online-softmax partials of the forward: 2 x G x ceil(G/64) x 4
    ws = torch.empty(2 * G + 4 + 8 * G * ((G + 63) // 64), device=a.device, dtype=torch.float32)
    tv, tdev = _temp_args(temp)
    check(lib().leccr_sim_infonce_fwd(ptr(a), ptr(b), ptr(idx), tv, G, D, int(both_dirs), ptr(loss),
                                      ptr(ws), tdev, stream()), 'leccr_sim_infonce_fwd')
    return loss, ws


def sim_infonce_bwd(a, b, idx, temp, ws, gscale, both_dirs=True, dtemp=None, rows=None):
    """rows = (row0, nrows): only those rows of da / db are computed (the rest is zero) and dtemp gets their share only."""
    G, D = a.shape
    r0, nr = rows if rows is not None else (0, G)
    da, db = (torch.empty_like(a), torch.empty_like(b)) if nr == G else (torch.zeros_like(a), torch.zeros_like(b))
    tv, tdev = _temp_args(temp)
    check(lib().leccr_sim_infonce_bwd(ptr(a), ptr(b), ptr(idx), tv, G, D, int(both_dirs), ptr(ws),
                                      float(gscale), ptr(da), ptr(db), ptr(dtemp), int(r0), int(nr), tdev, stream()),
          'leccr_sim_infonce_bwd')
    return da, db


def capsim_fwd(c, t, temp):
    b, n, D = _f32c(c).shape
    _f32c(t)
    loss = torch.empty(1, device=c.device, dtype=torch.float32)
    am = torch.empty(b, b, device=c.device, dtype=torch.int32)
    ws = torch.empty(b * b + 2 * b, device=c.device, dtype=torch.float32)
    tv, tdev = _temp_args(temp)
    check(lib().leccr_capsim_fwd(ptr(c), ptr(t), tv, b, n, D, ptr(loss), ptr(am), ptr(ws), tdev, stream()),
          'leccr_capsim_fwd')
    return loss, am, ws


def capsim_bwd(c, t, temp, am, ws, gscale, dtemp=None):
    b, n, D = c.shape
    dc, dt = torch.empty_like(c), torch.empty_like(t)
    tv, tdev = _temp_args(temp)
    check(lib().leccr_capsim_bwd(ptr(c), ptr(t), tv, b, n, D, ptr(am), ptr(ws), float(gscale), ptr(dc),
                                 ptr(dt), ptr(dtemp), tdev, stream()), 'leccr_capsim_bwd')
    return dc, dt


def dstl_fwd(v, c, s, t, alpha):
    G, n, D = _f32c(c).shape
    for x in (v, s, t):
        _f32c(x)
    loss = torch.empty(1, device=v.device, dtype=torch.float32)
    ws = torch.empty(3 * G * G + G + 8, device=v.device, dtype=torch.float32)
    check(lib().leccr_dstl_fwd(ptr(v), ptr(c), ptr(s), ptr(t), float(alpha), G, n, D, ptr(loss), ptr(ws), stream()),
          'leccr_dstl_fwd')
    return loss, ws


def dstl_bwd(v, t, ws, gscale, rows=None):
    G, D = v.shape
    r0, nr = rows if rows is not None else (0, G)
    dv, dt = (torch.empty_like(v), torch.empty_like(t)) if nr == G else (torch.zeros_like(v), torch.zeros_like(t))
    check(lib().leccr_dstl_bwd(ptr(v), ptr(t), G, D, ptr(ws), float(gscale), ptr(dv), ptr(dt), int(r0), int(nr), stream()),
          'leccr_dstl_bwd')
    return dv, dt


def cvpool_fwd(x):
    b, Lx, d = x.shape
    assert x.is_contiguous()
    pooled = torch.empty(b, d, device=x.device, dtype=torch.float32)
    inv = torch.empty_like(pooled)
    check(lib().leccr_cvpool_fwd(ptr(x), ptr(pooled), ptr(inv), b, Lx, d, dtype_code(x), stream()), 'leccr_cvpool_fwd')
    return pooled, inv


def cvpool_bwd(x, inv, dpooled):
    b, Lx, d = x.shape
    dx = torch.empty_like(x)
    check(lib().leccr_cvpool_bwd(ptr(x), ptr(inv), ptr(_f32c(dpooled)), ptr(dx), b, Lx, d, dtype_code(x), stream()),
          'leccr_cvpool_bwd')
    return dx


def capreg_fwd(c):
    b, n, d = c.shape
    assert c.is_contiguous()
    loss = torch.empty(1, device=c.device, dtype=torch.float32)
    check(lib().leccr_capreg_fwd(ptr(c), ptr(loss), b, n, d, dtype_code(c), stream()), 'leccr_capreg_fwd')
    return loss


def capreg_bwd(c, gscale):
    b, n, d = c.shape
    dc = torch.empty_like(c)
    check(lib().leccr_capreg_bwd(ptr(c), float(gscale), ptr(dc), b, n, d, dtype_code(c), stream()), 'leccr_capreg_bwd')
    return dc


# --------------------------------------------------------------------------------------- eval / optimizer
def rank_of_gt(scores, gt):
    """scores [R,C] f32; gt [R,per] int32 (-1 = unused) -> int32 [R] best rank of any ground truth."""
    R, Cn = _f32c(scores).shape
    assert gt.dtype == torch.int32 and gt.is_contiguous() and gt.shape[0] == R
    out = torch.empty(R, device=scores.device, dtype=torch.int32)
    check(lib().leccr_rank_of_gt(ptr(scores), ptr(gt), ptr(out), R, Cn, scores.stride(0), gt.shape[1], stream()),
          'leccr_rank_of_gt')
    return out


def adamw(p, g, m, v, seg_end, seg_wd, lr, beta1, beta2, eps, step, gscale=1.0, lp=None):
    """lp: optional bf16 tensor of p's size that receives the bf16 copy of the updated parameters in the same pass."""
    n = p.numel()
    if lp is not None:
        assert lp.dtype == torch.bfloat16 and lp.numel() == n and lp.is_contiguous()
    check(lib().leccr_adamw(ptr(p), ptr(g), ptr(m), ptr(v), n, ptr(seg_end), ptr(seg_wd), seg_end.numel(), float(lr),
                            float(beta1), float(beta2), float(eps), int(step), float(gscale), ptr(lp), stream()),
          'leccr_adamw')


def cast_bf16(src, dst):
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    check(lib().leccr_cast_bf16(ptr(src), ptr(dst), src.numel(), stream()), 'leccr_cast_bf16')
    return dst


def transpose_cast(src, dst):
    R, Cn = src.shape
    assert src.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and dst.numel() == R * Cn
    check(lib().leccr_transpose_cast(ptr(src), ptr(dst), R, Cn, dtype_code(dst), stream()), 'leccr_transpose_cast')
    return dst


def transpose_cast_batched(table, total_tiles):
    """table: int64 [n, 5] on the device (see leccr_hip.h)."""
    assert table.dtype == torch.int64 and table.dim() == 2 and table.shape[1] == 5 and table.is_contiguous()
    check(lib().leccr_transpose_cast_batched(ptr(table), table.shape[0], int(total_tiles), stream()),
          'leccr_transpose_cast_batched')


def split3_weights_batched(table, total_tiles):
    """Both split copies (G form [R, 3 C]; X form of the transpose [C, 3 R]) of every weight listed in `table` (int64 [n, 6] on the device:
    first tile, src, dstG, dstT, R, C) in one launch: the fp32 path's counterpart of transpose_cast_batched (leccr_hip.h)."""
    check(lib().leccr_split3_weights_batched(ptr(table), table.shape[0], int(total_tiles), stream()), 'leccr_split3_weights_batched')


def fp8_requantize_weights(holders):
    """All fp8 weight copies in three launches (leccr_fp8_requantize_weights): holders carry _lp (bf16 [R, C]), _q8 ([R, C]), _q8t ([C, R])
    and _wsite.  Returns a closure that re-runs the batch (the table lives on the device, built once)."""
    rows, tiles = [], 0
    for h in holders:
        R, Cn = h._lp.shape
        assert h._lp.is_contiguous() and h._q8.is_contiguous() and h._q8t.is_contiguous() and R % 64 == 0 and Cn % 64 == 0
        assert h._q8.shape == (R, Cn) and h._q8t.shape == (Cn, R)
        rows.append([tiles, h._lp.data_ptr(), h._q8.data_ptr(), h._q8t.data_ptr(), R, Cn, h._wsite.row.data_ptr()])
        tiles += (R // 64) * (Cn // 64)
    table = torch.tensor(rows, dtype=torch.int64, device=holders[0]._lp.device)

    def run():
        check(lib().leccr_fp8_requantize_weights(ptr(table), table.shape[0], tiles, stream()), 'leccr_fp8_requantize_weights')
        for h in holders:
            h._wsite.primed = True
    return run


def axpy(x, y, a):
    assert x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    check(lib().leccr_axpy(ptr(x), ptr(y), float(a), x.numel(), dtype_code(x), stream()), 'leccr_axpy')
    return y


def dropout(x, p, seed, out=None):
    """y = keep(seed, i) ? x / (1-p) : 0; the same call on a gradient is the backward.  out may be x (in place)."""
    assert x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    check(lib().leccr_dropout(ptr(x), ptr(out), x.numel(), float(p), int(seed), dtype_code(x), stream()), 'leccr_dropout')
    return out


# --------------------------------------------------------------------------------------- fp8 GEMM operands
class Fp8Sites:
    """A block of quantisation sites ({amax, scale, 1/scale, format} x n, one f32 tensor on the device; see leccr_hip.h).
    new(fmt) hands out one site; update() turns the amax every site has seen into its next scale (once per step)."""

    E4M3, E5M2 = 0, 1

    def __init__(self, device, capacity=4096):
        self.state = torch.zeros(capacity, 4, device=device, dtype=torch.float32)
        self.state[:, 1:3] = 1.0
        self.n = 0

    def new(self, fmt):
        assert self.n < self.state.shape[0]
        i, self.n = self.n, self.n + 1
        self.state[i, 3] = float(fmt)
        return Fp8Site(self.state[i], fmt)

    # share of a site's amax that the next step's running maximum starts from (leccr_fp8_update_keep; LECCR_FP8_AMAX_KEEP=0: from zero)
    keep = float(os.environ.get('LECCR_FP8_AMAX_KEEP', '0.9'))

    def update(self, keep=None):
        if self.n:
            check(lib().leccr_fp8_update_keep(ptr(self.state), self.n, self.keep if keep is None else keep, stream()), 'leccr_fp8_update_keep')


class Fp8Site:
    def __init__(self, row, fmt):
        self.row, self.fmt, self.primed = row, fmt, False
        self.inv_scale = row[2:3]                                    # device scalar handed to leccr_gemm (alpha_dev)
        self.torch_dtype = torch.float8_e4m3fn if fmt == Fp8Sites.E4M3 else torch.float8_e5m2

    def quantize(self, x, out=None):
        """x [rows, cols] (bf16 or f32, unit inner stride) -> its fp8 copy under this site's scale.  The first use scales from the
        tensor itself (amax pass + update); afterwards the scale is the one the previous step's amax gave (delayed scaling)."""
        assert x.dim() == 2 and x.stride(1) == 1
        rows, cols = x.shape
        dt = torch.float8_e4m3fn if self.fmt == Fp8Sites.E4M3 else torch.float8_e5m2
        y = torch.empty(rows, cols, device=x.device, dtype=dt) if out is None else out
        assert y.shape == (rows, cols) and y.stride(1) == 1 and y.dtype == dt
        if not self.primed:
            check(lib().leccr_fp8_amax(ptr(x), rows, cols, x.stride(0), dtype_code(x), ptr(self.row), stream()), 'leccr_fp8_amax')
            check(lib().leccr_fp8_update(ptr(self.row), 1, stream()), 'leccr_fp8_update')
            # A site can be shared by passes on DIFFERENT streams (the caption encoder IS the text encoder: its forward runs on the caption
            # stream beside the text pass on the text stream).  `primed` is a host flag: the second pass would go on to read a scale that the
            # first pass's update kernel, on another stream, may not have written yet - which scale it found decided the first step's
            # quantisation and, through the gradient sites' first amax, now and then a power of two of a later scale (round 5: the fp8
            # A/B test differed from run to run with the order of the tests before it).  Priming happens once per site, in the first
            # step: the host waits for it, so every launch enqueued afterwards - on any stream - sees the scale.
            torch.cuda.current_stream(x.device).synchronize()
            self.primed = True
        check(lib().leccr_fp8_quantize(ptr(x), ptr(y), rows, cols, x.stride(0), y.stride(0), dtype_code(x), self.fmt, ptr(self.row),
                                       stream()), 'leccr_fp8_quantize')
        return y

    def quantize_current(self, x, outs):
        """CURRENT scaling (weights): the scale is derived from x itself right now - amax pass, update of this one site, then the
        quantise passes of x's layouts `outs` = [(tensor, fp8 out), ..] - so the fp8 copies and the 1/scale the GEMMs read at kernel
        time always belong together.  (Delayed scaling is wrong for a tensor that is quantised ONCE and consumed across a later
        update(): the copy would carry the old scale and the product the new one - off by 2x whenever amax crosses a power of two.)"""
        rows, cols = x.shape
        check(lib().leccr_fp8_amax(ptr(x), rows, cols, x.stride(0), dtype_code(x), ptr(self.row), stream()), 'leccr_fp8_amax')
        check(lib().leccr_fp8_update(ptr(self.row), 1, stream()), 'leccr_fp8_update')
        self.primed = True
        for t, y in outs:
            self.quantize(t, out=y)
