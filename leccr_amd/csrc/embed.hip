// embed.hip - input stages of the two towers (all HBM-bound, 16-byte accesses where the geometry allows):
//   patchify        image NCHW f32 -> patch rows (conv1 with stride == kernel is a GEMM over these rows)
//   vit_assemble    [CLS] + patch tokens + positional embedding, and its backward
//   bert_embed      word/position/token-type gather-sum, and the scatter-add backward
#include "common.h"

namespace leccr {

// one thread per 4 consecutive output columns (p % 4 == 0) or per column (VEC = 1)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int64_t b,
                                                       int R, int p, int g, int Kp) {
    const int64_t total = b * g * g * (int64_t)(Kp / VEC);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(t % (Kp / VEC)) * VEC;
        const int64_t row = t / (Kp / VEC);
        const int gx = (int)(row % g), gy = (int)((row / g) % g);
        const int64_t bi = row / ((int64_t)g * g);
        T* o = out + row * Kp + col;
        if (col >= 3 * p * p) {
            for (int j = 0; j < VEC; ++j) o[j] = from_f32<T>(0.f);
            continue;
        }
        const int c = col / (p * p), py = (col / p) % p, px = col % p;
        const float* src = img + ((bi * 3 + c) * R + (int64_t)gy * p + py) * R + (int64_t)gx * p + px;
        if constexpr (VEC == 4) {
            store4(o, *reinterpret_cast<const f32x4*>(src));
        } else {
            o[0] = from_f32<T>(src[0]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_fwd_kernel(const T* __restrict__ patch_out,
                                                               const float* __restrict__ cls,
                                                               const float* __restrict__ pos, T* __restrict__ tok,
                                                               int64_t b, int N, int d) {
    const int dq = d / 4;
    const int64_t total = b * N * (int64_t)dq;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % dq) * 4;
        const int64_t r = t / dq;
        const int n = (int)(r % N);
        const int64_t bi = r / N;
        f32x4 v = load4(pos + (int64_t)n * d + c);
        if (n == 0) v += load4(cls + c);
        else v += load4(patch_out + (bi * (N - 1) + n - 1) * d + c);
        store4(tok + r * d + c, v);
    }
}

// thread per (n, 4 columns): copies the patch rows' gradient, reduces over the batch for dpos / dcls
template <typename T>
__global__ __launch_bounds__(256) void vit_assemble_bwd_kernel(const T* __restrict__ dtok, T* __restrict__ dpatch,
                                                               float* __restrict__ dcls, float* __restrict__ dpos,
                                                               int64_t b, int N, int d) {
    const int dq = d / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (int64_t)N * dq) return;
    const int c = (int)(t % dq) * 4, n = (int)(t / dq);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t bi = 0; bi < b; ++bi) {
        const f32x4 g = load4(dtok + (bi * N + n) * d + c);
        acc += g;
        if (n > 0) store4(dpatch + (bi * (N - 1) + n - 1) * d + c, g);
    }
    float* dp = dpos + (int64_t)n * d + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) dp[j] += acc[j];
    if (n == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dcls[c + j] += acc[j];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const int64_t* __restrict__ ids,
                                                             const float* __restrict__ word,
                                                             const float* __restrict__ pos,
                                                             const float* __restrict__ type0, T* __restrict__ emb,
                                                             int64_t rows, int L, int d, int64_t vocab) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
    const int l = (int)(row % L);
    for (int c = lane * 4; c < d; c += 256)
        store4(emb + row * d + c, load4(word + id * d + c) + load4(pos + (int64_t)l * d + c) + load4(type0 + c));
}

template <typename T>
__global__ __launch_bounds__(256) void bert_embed_scatter_kernel(const int64_t* __restrict__ ids,
                                                                 const T* __restrict__ demb,
                                                                 float* __restrict__ dword, int64_t rows, int d,
                                                                 int64_t vocab, int64_t pad_id) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t id = ids[row];
    if (id == pad_id || id < 0 || id >= vocab) return;  // nn.Embedding(padding_idx): no gradient for the pad row
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 g = load4(demb + row * d + c);
        float* dst = dword + id * d + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(dst + j, g[j]);
    }
}

static unsigned grid_for(int64_t total) {
    int64_t nb = cdiv(total, 256);
    return (unsigned)(nb > 8192 ? 8192 : (nb < 1 ? 1 : nb));
}

}  // namespace leccr

using namespace leccr;

extern "C" int leccr_patchify(const float* image, void* patches, int64_t b, int64_t R, int64_t p, int64_t Kp,
                              int dtype, void* stream) {
    LECCR_REQUIRE(image && patches, LECCR_EINVAL, "patchify: null pointer");
    LECCR_REQUIRE(p > 0 && R >= p && Kp >= 3 * p * p && (Kp & 3) == 0, LECCR_EINVAL, "patchify: bad geometry");    // R % p pixels at the right / bottom edge are not covered, as with the stride-p convolution
    if (b == 0) return LECCR_OK;
    const int g = (int)(R / p);
    hipStream_t st = as_stream(stream);
    const bool vec = (p % 4 == 0) && (R % 4 == 0);
    const int64_t total = b * g * g * (vec ? Kp / 4 : Kp);
#define LP(T, V) hipLaunchKernelGGL((patchify_kernel<T, V>), dim3(grid_for(total)), dim3(256), 0, st, image, (T*)patches, b, (int)R, (int)p, g, (int)Kp)
    if (dtype == LECCR_F32) { if (vec) LP(float, 4); else LP(float, 1); }
    else if (dtype == LECCR_BF16) { if (vec) LP(bf16_t, 4); else LP(bf16_t, 1); }
    else { set_error("patchify: bad dtype"); return LECCR_EINVAL; }
#undef LP
    return check_launch("patchify");
}

extern "C" int leccr_vit_assemble_fwd(const void* patch_out, const float* cls, const float* pos, void* tok, int64_t b,
                                      int64_t N, int64_t d, int dtype, void* stream) {
    LECCR_REQUIRE(patch_out && cls && pos && tok && N >= 1 && d > 0 && (d & 3) == 0, LECCR_EINVAL, "vit_assemble_fwd: bad argument");
    if (b == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    const int64_t total = b * N * (d / 4);
    if (dtype == LECCR_F32)
        hipLaunchKernelGGL((vit_assemble_fwd_kernel<float>), dim3(grid_for(total)), dim3(256), 0, st, (const float*)patch_out, cls, pos, (float*)tok, b, (int)N, (int)d);
    else if (dtype == LECCR_BF16)
        hipLaunchKernelGGL((vit_assemble_fwd_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)patch_out, cls, pos, (bf16_t*)tok, b, (int)N, (int)d);
    else { set_error("vit_assemble_fwd: bad dtype"); return LECCR_EINVAL; }
    return check_launch("vit_assemble_fwd");
}

extern "C" int leccr_vit_assemble_bwd(const void* dtok, void* dpatch_out, float* dcls, float* dpos, int64_t b, int64_t N,
                                      int64_t d, int dtype, void* stream) {
    LECCR_REQUIRE(dtok && dpatch_out && dcls && dpos && N >= 1 && d > 0 && (d & 3) == 0, LECCR_EINVAL, "vit_assemble_bwd: bad argument");
    if (b == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)cdiv(N * (d / 4), 256);
    if (dtype == LECCR_F32)
        hipLaunchKernelGGL((vit_assemble_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)dtok, (float*)dpatch_out, dcls, dpos, b, (int)N, (int)d);
    else if (dtype == LECCR_BF16)
        hipLaunchKernelGGL((vit_assemble_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dtok, (bf16_t*)dpatch_out, dcls, dpos, b, (int)N, (int)d);
    else { set_error("vit_assemble_bwd: bad dtype"); return LECCR_EINVAL; }
    return check_launch("vit_assemble_bwd");
}

extern "C" int leccr_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0,
                                    void* emb, int64_t b, int64_t L, int64_t d, int64_t vocab, int dtype, void* stream) {
    LECCR_REQUIRE(ids && word && pos && type0 && emb && L > 0 && d > 0 && (d & 3) == 0 && vocab > 0, LECCR_EINVAL, "bert_embed_fwd: bad argument");
    if (b == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)cdiv(b * L, 4);
    ProfScope prof(PROF_EMBED, (double)b * L * (8.0 + d * (4.0 + (dtype == LECCR_BF16 ? 2.0 : 4.0))), st);   // word row in, emb row out
    if (dtype == LECCR_F32)
        hipLaunchKernelGGL((bert_embed_fwd_kernel<float>), dim3(grid), dim3(256), 0, st, ids, word, pos, type0, (float*)emb, b * L, (int)L, (int)d, vocab);
    else if (dtype == LECCR_BF16)
        hipLaunchKernelGGL((bert_embed_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, ids, word, pos, type0, (bf16_t*)emb, b * L, (int)L, (int)d, vocab);
    else { set_error("bert_embed_fwd: bad dtype"); return LECCR_EINVAL; }
    return check_launch("bert_embed_fwd");
}

extern "C" int leccr_bert_embed_bwd(const int64_t* ids, const void* demb, float* dword, float* dpos, float* dtype0,
                                    int64_t b, int64_t L, int64_t d, int64_t vocab, int64_t pad_id, int dtype,
                                    void* stream) {
    LECCR_REQUIRE(ids && demb && dword && dpos && dtype0 && L > 0 && d > 0 && (d & 3) == 0, LECCR_EINVAL, "bert_embed_bwd: bad argument");
    if (b == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    const unsigned grid = (unsigned)cdiv(b * L, 4);
    if (dtype == LECCR_F32)
        hipLaunchKernelGGL((bert_embed_scatter_kernel<float>), dim3(grid), dim3(256), 0, st, ids, (const float*)demb, dword, b * L, (int)d, vocab, pad_id);
    else if (dtype == LECCR_BF16)
        hipLaunchKernelGGL((bert_embed_scatter_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, ids, (const bf16_t*)demb, dword, b * L, (int)d, vocab, pad_id);
    else { set_error("bert_embed_bwd: bad dtype"); return LECCR_EINVAL; }
    if (int rc = check_launch("bert_embed_scatter")) return rc;
    // dpos[l, :] += sum_b demb[b, l, :]  == column sums of demb viewed as [b, L*d]
    if (int rc = leccr_colsum(demb, dpos, b, L * d, L * d, dtype, stream)) return rc;
    // dtype0[:] += sum_{b,l} demb[b, l, :]
    return leccr_colsum(demb, dtype0, b * L, d, d, dtype, stream);
}
