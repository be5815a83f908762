"""ctypes binding of libleccr_hip.so (the C ABI declared in include/leccr_hip.h).

No fallback: if the library is missing or a call fails this raises - the product path never silently runs
PyTorch eager or CPU code in its place.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LECCR_LIB') or os.path.join(_HERE, 'libleccr_hip.so')      # LECCR_LIB: a diagnostic build (scripts/build_variant.sh)

F32, BF16, FP8, FP8_BF8A = 0, 1, 2, 3      # FP8 / FP8_BF8A: leccr_gemm operand codes only (e4m3 x e4m3 / A e5m2 x B e4m3)
EPI_NONE, EPI_QUICKGELU, EPI_GELU, EPI_DQUICKGELU, EPI_DGELU = 0, 1, 2, 3, 4
EPI_QUICKGELU_D, EPI_GELU_D, EPI_MULAUX = 5, 6, 7      # the activation's DERIVATIVE saved in aux_out / multiplied in by the backward epilogue

vp, i64, i32, f32 = C.c_void_p, C.c_int64, C.c_int32, C.c_float


class GemmParams(C.Structure):
    _fields_ = [('A', vp), ('B', vp), ('C', vp), ('M', i64), ('N', i64), ('K', i64), ('lda', i64), ('ldb', i64),
                ('ldc', i64), ('ab_dtype', i32), ('c_dtype', i32), ('trans_a', i32), ('trans_b', i32), ('bias', vp),
                ('epilogue', i32), ('accumulate', i32), ('aux', vp), ('ldaux', i64), ('aux_out', vp),
                ('residual', vp), ('ldr', i64), ('alpha', f32), ('reserved', i32), ('colsum', vp), ('asum', vp), ('alpha_dev', vp * 2), ('c8', vp), ('ldc8', i64), ('c8_site', vp), ('c8_fmt', i32),
                ('reserved2', i32), ('c3', vp), ('c3_form', i32), ('asum_x3', i32)]


_SIGS = {
    'leccr_version': ([], C.c_int),
    'leccr_last_error': ([], C.c_char_p),
    'leccr_set_force_generic': ([C.c_int], C.c_int),
    'leccr_set_gemm_variant': ([C.c_int], C.c_int),
    'leccr_last_gemm_kernel': ([], C.c_int),
    'leccr_set_gemm_max_blocks': ([C.c_int], C.c_int),
    'leccr_set_gemm_max_blocks_for': ([C.c_int, C.c_int64], C.c_int),
    'leccr_get_gemm_max_blocks': ([], C.c_int),
    'leccr_prof_enable': ([C.c_int], C.c_int),
    'leccr_prof_read': ([C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)], C.c_int),
    'leccr_prof_read_n': ([C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)], C.c_int),
    'leccr_prof_read_union': ([C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)], C.c_int),
    'leccr_prof_dropped': ([], C.c_int64),
    'leccr_set_attn_variant': ([C.c_int], C.c_int),
    'leccr_last_attn_kernel': ([], C.c_int),
    'leccr_gemm': ([C.POINTER(GemmParams), vp], C.c_int),
    'leccr_split3': ([vp, vp, i64, i64, i64, C.c_int, C.c_int, vp, vp], C.c_int),
    'leccr_split_n': ([vp, vp, i64, i64, i64, C.c_int, C.c_int, C.c_int, vp, vp], C.c_int),
    'leccr_debug_set_stamps': ([vp], C.c_int),
    'leccr_debug_occupy': ([C.c_int, C.c_int, C.c_int64, vp], C.c_int),
    'leccr_layernorm_fwd': ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, C.c_int, vp], C.c_int),
    'leccr_layernorm_bwd': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, i64, i64, C.c_int, vp], C.c_int),
    'leccr_layernorm_fwd_drop': ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, f32, C.c_uint64, C.c_int, vp], C.c_int),
    'leccr_layernorm_bwd_drop': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, C.c_uint64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_layernorm_fwd_s3': ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, f32, C.c_uint64, vp, C.c_int, vp], C.c_int),
    'leccr_layernorm_fwd_ex': ([vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, f32, f32, C.c_uint64, vp, vp, C.c_int, C.c_int, vp], C.c_int),
    'leccr_layernorm_bwd_ex': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, C.c_uint64, vp, vp, C.c_int, i64, i64, C.c_int, vp], C.c_int),
    'leccr_attn_fwd': ([vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, C.c_int, f32, C.c_uint64, vp], C.c_int),
    'leccr_attn_bwd': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64,
                        C.c_int, f32, C.c_uint64, vp], C.c_int),
    'leccr_attn_fwd_causal': ([vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_attn_bwd_causal': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_attn_fwd_q8': ([vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64, C.c_int, f32, C.c_uint64, vp, i64, vp, vp], C.c_int),
    'leccr_attn_bwd_q8': ([vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, i64, i64, i64, i64,
                           C.c_int, f32, C.c_uint64, vp, vp, vp, i64, vp, vp], C.c_int),
    'leccr_attn_s3_takes': ([i64] * 10, C.c_int),
    'leccr_attn_fwd_s3': ([vp] * 6 + [i64] * 10 + [f32, C.c_uint64, vp], C.c_int),
    'leccr_attn_bwd_s3': ([vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp] + [i64] * 11 + [vp, f32, C.c_uint64, vp], C.c_int),
    'leccr_patchify': ([vp, vp, i64, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_vit_assemble_fwd': ([vp, vp, vp, vp, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_vit_assemble_bwd': ([vp, vp, vp, vp, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_bert_embed_fwd': ([vp, vp, vp, vp, vp, i64, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_bert_embed_bwd': ([vp, vp, vp, vp, vp, i64, i64, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_colsum': ([vp, vp, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_fp8_amax': ([vp, i64, i64, i64, C.c_int, vp, vp], C.c_int),
    'leccr_fp8_update': ([vp, i64, vp], C.c_int),
    'leccr_fp8_update_keep': ([vp, i64, f32, vp], C.c_int),
    'leccr_fp8_quantize': ([vp, vp, i64, i64, i64, i64, C.c_int, C.c_int, vp, vp], C.c_int),
    'leccr_fp8_requantize_weights': ([vp, i64, i64, vp], C.c_int),
    'leccr_l2norm_fwd': ([vp, vp, vp, i64, i64, C.c_int, vp], C.c_int),
    'leccr_l2norm_bwd': ([vp, vp, vp, vp, i64, i64, C.c_int, vp], C.c_int),
    'leccr_sim_infonce_fwd': ([vp, vp, vp, f32, i64, i64, C.c_int, vp, vp, vp, vp], C.c_int),
    'leccr_sim_infonce_bwd': ([vp, vp, vp, f32, i64, i64, C.c_int, vp, f32, vp, vp, vp, i64, i64, vp, vp], C.c_int),
    'leccr_capsim_fwd': ([vp, vp, f32, i64, i64, i64, vp, vp, vp, vp, vp], C.c_int),
    'leccr_capsim_bwd': ([vp, vp, f32, i64, i64, i64, vp, vp, f32, vp, vp, vp, vp, vp], C.c_int),
    'leccr_dstl_fwd': ([vp, vp, vp, vp, f32, i64, i64, i64, vp, vp, vp], C.c_int),
    'leccr_dstl_bwd': ([vp, vp, i64, i64, vp, f32, vp, vp, i64, i64, vp], C.c_int),
    'leccr_cvpool_fwd': ([vp, vp, vp, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_cvpool_bwd': ([vp, vp, vp, vp, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_capreg_fwd': ([vp, vp, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_capreg_bwd': ([vp, f32, vp, i64, i64, i64, C.c_int, vp], C.c_int),
    'leccr_rank_of_gt': ([vp, vp, vp, i64, i64, i64, i64, vp], C.c_int),
    'leccr_adamw': ([vp, vp, vp, vp, i64, vp, vp, i64, f32, f32, f32, f32, i64, f32, vp, vp], C.c_int),
    'leccr_cast_bf16': ([vp, vp, i64, vp], C.c_int),
    'leccr_transpose_cast': ([vp, vp, i64, i64, C.c_int, vp], C.c_int),
    'leccr_transpose_cast_batched': ([vp, i64, i64, vp], C.c_int),
    'leccr_split3_weights_batched': ([vp, i64, i64, vp], C.c_int),
    'leccr_axpy': ([vp, vp, f32, i64, C.c_int, vp], C.c_int),
    'leccr_dropout': ([vp, vp, i64, f32, C.c_uint64, C.c_int, vp], C.c_int),
}

_lib = None


class LeccrHipError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LeccrHipError(f'{LIB_PATH} is missing: build it with `python -m leccr_amd.build` '
                                '(there is no fallback path)')
        l = C.CDLL(LIB_PATH)
        for name, (argtypes, restype) in _SIGS.items():
            fn = getattr(l, name)        # AttributeError if the library does not export a declared symbol
            fn.argtypes, fn.restype = argtypes, restype
        _lib = l
    return _lib


def exported_symbols():
    return list(_SIGS)


def check(rc, what):
    if rc != 0:
        raise LeccrHipError(f'{what} failed (rc={rc}): {lib().leccr_last_error().decode()}')


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise LeccrHipError(f'unsupported dtype {t.dtype}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must live on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise LeccrHipError('leccr_amd ops need GPU tensors (no CPU fallback)')
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
