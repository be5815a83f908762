// norm.hip - LayerNorm forward/backward, row L2-normalise forward/backward, column sums (bias gradients).
// All HBM-bound: one wave64 per row (bf16: one or two rows with 16-byte accesses where the width allows), statistics in f32,
// wave-shuffle reductions (no LDS for the row statistics).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace leccr {

// Optional second output of the LayerNorm passes: the fp8 copy of the tensor just written, for the scaled-MFMA product that consumes it
// (compute_dtype 'fp8').  Same arithmetic as leccr_fp8_quantize on the stored tensor: the value is rounded through T first, scaled by
// site[1], clamped and converted; |value| feeds the site's running amax (one conditional atomic per wave).
struct Q8Out {
    uint8_t* out;      // [rows, d] fp8, or nullptr: off
    float* site;       // {amax, scale, 1/scale, fmt}
    int fmt;           // 0 e4m3, 1 e5m2; 2 / 3 (f32 rows only, no site): `out` is the bf16 (hi, lo) split copy [rows, 3 d] of leccr_split3 in
                       // form 0 (hi|hi|lo) / 1 (hi|lo|hi) - the operand copy of the fp32 path's next product (leccr_layernorm_fwd_s3)
};
__device__ __forceinline__ int q8_pack4(const Q8Out& q, float sc, float a, float b, float c, float d) {
    const float fmax = q.fmt ? 57344.f : 448.f;
    a = fminf(fmaxf(a * sc, -fmax), fmax);
    b = fminf(fmaxf(b * sc, -fmax), fmax);
    c = fminf(fmaxf(c * sc, -fmax), fmax);
    d = fminf(fmaxf(d * sc, -fmax), fmax);
    int w = 0;
    if (q.fmt == 0) {
        w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    } else {
        w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
        w = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
    }
    return w;
}
__device__ __forceinline__ void q8_amax(const Q8Out& q, float m) {
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > __builtin_nontemporal_load(q.site))
        atomicMax(reinterpret_cast<unsigned int*>(q.site), __float_as_uint(m));      // m >= 0: float order == unsigned order of the bits
}

// A wave owns one row; lane l handles the 4-element chunks c = l, l+64, ... (NCH chunks per lane, d <= NCH*256).
// EX: the instantiation with the optional extras (dropout on x, fp8 copy of y); the plain one carries none of their code.
template <typename T, int NCH, bool EX>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ add,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            T* __restrict__ sum_out, float* __restrict__ mean,
                                                            float* __restrict__ rstd, int64_t rows, int d, float eps,
                                                            DropCfg drop, Q8Out q8) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * d;
    f32x4 v[NCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            v[i] = load4(xr + c);
            if constexpr (EX) {
                if (drop.thr) {     // x = dropout(x): the mask leccr_dropout(x, seed) applies, rounded through T as its output would be
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        v[i][j] = to_f32(from_f32<T>(v[i][j] * drop_mul(drop, (uint32_t)(row * d + c + j))));
                }
            }
            if (add) v[i] += load4(add + row * d + c);
            if (sum_out) {
                // normalise exactly what is stored (and later handed to the backward as x_in): round through T
                store4(sum_out + row * d + c, v[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[i][j] = to_f32(from_f32<T>(v[i][j]));
            }
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        } else {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float mu = wave_sum(s) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const f32x4 t = v[i] - mu;
            q += t[0] * t[0] + t[1] * t[1] + t[2] * t[2] + t[3] * t[3];
        }
    }
    const float rs = rsqrtf(wave_sum(q) / d + eps);
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
    [[maybe_unused]] float q8sc = 1.f, q8m = 0.f;
    if constexpr (EX) q8sc = (q8.out && q8.fmt < 2) ? q8.site[1] : 1.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < d) {
            const f32x4 g = load4(gamma + c), b = load4(beta + c);
            const f32x4 o = (v[i] - mu) * rs * g + b;
            if (y) store4(y + row * d + c, o);
            if (EX && q8.out && q8.fmt >= 2) {       // the split copy of the fp32 path (a wave covers 256 consecutive columns per segment)
                bf16x4 h4, l4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h4[j] = (bf16_t)to_f32(from_f32<T>(o[j]));
                    l4[j] = (bf16_t)(to_f32(from_f32<T>(o[j])) - (float)h4[j]);
                }
                bf16_t* d3 = reinterpret_cast<bf16_t*>(q8.out) + row * 3 * d + c;
                *reinterpret_cast<bf16x4*>(d3) = h4;
                *reinterpret_cast<bf16x4*>(d3 + d) = q8.fmt == 2 ? h4 : l4;
                *reinterpret_cast<bf16x4*>(d3 + 2 * d) = q8.fmt == 2 ? l4 : h4;
            } else if (EX && q8.out) {
                float r[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    r[j] = to_f32(from_f32<T>(o[j]));
                    q8m = fmaxf(q8m, fabsf(r[j]));
                }
                *reinterpret_cast<int*>(q8.out + row * d + c) = q8_pack4(q8, q8sc, r[0], r[1], r[2], r[3]);
            }
        }
    }
    if (EX && q8.out && q8.fmt < 2) q8_amax(q8, q8m);
}

// Backward.  Grid-stride over rows so each wave accumulates its dgamma/dbeta partials in registers; one LDS reduce
// per block and one f32 atomic per (block, column).  gamma is re-read per row (L1-resident) and the colsum partials exist only
// in the CS instantiation: the kernel is bound by how many rows are in flight, i.e. by waves per SIMD (94 registers: 5 waves;
// 106 with the colsum partials: 4 waves - forcing 5 there spills and costs 75 %).
// EX: the instantiation with the optional second outputs (dropout(dx), fp8 copy); the plain one keeps its 94 registers / 5 waves.
template <typename T, int NCH, bool CS, bool EX>
__global__ __launch_bounds__(256, NCH <= 3 ? ((CS || EX) ? 4 : 5) : (NCH == 4 ? 4 : 1)) void layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ xin,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const T* __restrict__ dres,
                                                            T* __restrict__ dx, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ colsum,
                                                            int colsum_of, int64_t rows, int d, T* __restrict__ dx_drop,
                                                            DropCfg drop, Q8Out q8) {
    __shared__ float red[NCH * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[NCH], db[NCH], dc[CS ? NCH : 1];    // dc: column sums of dres (colsum_of = 1) or of the dx written (2)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (CS) dc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    [[maybe_unused]] float q8sc = 1.f, q8m = 0.f;      // fp8 copy of the gradient the dense layer sees (dx_drop if given, else dx)
    if constexpr (EX) q8sc = (q8.out && q8.fmt < 2) ? q8.site[1] : 1.f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[NCH], gy[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                const f32x4 dyv = load4(dy + row * d + c);
                xh[i] = (load4(xin + row * d + c) - mu) * rs;
                gy[i] = dyv * load4(gamma + c);
                dg[i] += dyv * xh[i];
                db[i] += dyv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 += gy[i][j];
                    s2 += gy[i][j] * xh[i][j];
                }
            }
        }
        s1 = wave_sum(s1) / d;
        s2 = wave_sum(s2) / d;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                f32x4 o = (gy[i] - s1 - xh[i] * s2) * rs;
                if (dres) {
                    const f32x4 r = load4(dres + row * d + c);     // (prefetching it with dy / x costs occupancy: measured slower)
                    o += r;
                    if constexpr (CS) {
                        if (colsum_of == 1) dc[i] += r;
                    }
                }
                if constexpr (CS) {
                    if (colsum_of == 2) dc[i] += o;
                }
                store4(dx + row * d + c, o);
                if constexpr (EX) {
                    f32x4 od = o;
                    if (dx_drop) {      // second output: dropout(dx) with the forward's mask - what leccr_dropout(dx, seed) would write
#pragma unroll
                        for (int j = 0; j < 4; ++j) od[j] = to_f32(from_f32<T>(o[j])) * drop_mul(drop, (uint32_t)(row * d + c + j));
                        store4(dx_drop + row * d + c, od);
                    }
                    if (q8.out && q8.fmt >= 2) {       // the fp32 path's split copy of that gradient (f32 rows; fmt 2: hi | hi | lo, 3: hi | lo | hi)
                        bf16x4 h4, l4;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            h4[j] = (bf16_t)od[j];
                            l4[j] = (bf16_t)(od[j] - (float)h4[j]);
                        }
                        bf16_t* d3 = reinterpret_cast<bf16_t*>(q8.out) + row * 3 * d + c;
                        *reinterpret_cast<bf16x4*>(d3) = h4;
                        *reinterpret_cast<bf16x4*>(d3 + d) = q8.fmt == 2 ? h4 : l4;
                        *reinterpret_cast<bf16x4*>(d3 + 2 * d) = q8.fmt == 2 ? l4 : h4;
                    } else if (q8.out) {
                        float r[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            r[j] = to_f32(from_f32<T>(od[j]));
                            q8m = fmaxf(q8m, fabsf(r[j]));
                        }
                        *reinterpret_cast<int*>(q8.out + row * d + c) = q8_pack4(q8, q8sc, r[0], r[1], r[2], r[3]);
                    }
                }
            }
        }
    }
    if constexpr (EX) {
        if (q8.out && q8.fmt < 2) q8_amax(q8, q8m);
    }
    // reduce the 4 waves' partials through LDS (one wave at a time), then one f32 atomic per column
    for (int pass = 0; pass < (CS ? 3 : 2); ++pass) {
        for (int w = 0; w < 4; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + 64 * i) * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float val = pass == 0 ? dg[i][j] : pass == 1 ? db[i][j] : dc[CS ? i : 0][j];
                        red[c + j] = w == 0 ? val : red[c + j] + val;
                    }
                }
            }
        }
        __syncthreads();
        float* out = pass == 0 ? dgamma : pass == 1 ? dbeta : colsum;
        for (int c = threadIdx.x; c < d; c += 256) atomicAdd(out + c, red[c]);
    }
}

// Backward, bf16 rows, SOFTWARE-PIPELINED over the rows of a wave (round 5).  The kernel above runs one row per wave at a time: load dy and x
// (a memory latency), reduce, load the residual gradient (a second latency), store - with 5 waves per SIMD the chip has 5120 rows in flight and the
// launch on [147712, 768] is bound by those two latencies per row (188 us = 4.8 TB/s), not by bytes.  Here a wave issues the NEXT row's dy / x loads
// and THIS row's residual-gradient load before it touches the current row's values (packed bf16: 18 more registers, 4 waves per SIMD instead of 5,
// but 2.5 rows in flight per wave instead of 1).  Same arithmetic, same order of the dgamma / dbeta sums per wave: results identical to the plain kernel.
// EX: the second outputs of the plain kernel's EX instantiation (dropout(dx), the fp8 copy of the gradient the dense layer sees); NCH = 4: the 1024-wide rows
// of ViT-L / XLM-R-large (3 waves per SIMD there).
template <int NCH, bool EX>
__global__ __launch_bounds__(256, (NCH <= 3 && !EX) ? 4 : 3) void layernorm_bwd_pipe_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ xin, const float* __restrict__ gamma,
                                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                    const bf16_t* __restrict__ dres, bf16_t* __restrict__ dx, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int64_t rows, int d, bf16_t* __restrict__ dx_drop, DropCfg drop, Q8Out q8) {
    __shared__ float red[NCH * 256];
    [[maybe_unused]] float q8sc = 1.f, q8m = 0.f;
    if constexpr (EX) q8sc = q8.out ? q8.site[1] : 1.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 dg[NCH], db[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        dg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t row = (int64_t)blockIdx.x * 4 + wave;
    bf16x4 dyr[NCH], xr[NCH];
    float mu = 0.f, rs = 0.f;
    auto cvt = [](bf16x4 v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; };
    if (row < rows) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                dyr[i] = *reinterpret_cast<const bf16x4*>(dy + row * d + c);
                xr[i] = *reinterpret_cast<const bf16x4*>(xin + row * d + c);
            }
        }
        mu = mean[row];
        rs = rstd[row];
    }
    while (row < rows) {
        const int64_t nxt = row + stride;
        bf16x4 dyn[NCH], xn[NCH], rr[NCH];
        float mun = 0.f, rsn = 0.f;
        // in flight while this row is worked on: its residual gradient, and the next row's dy / x
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                if (dres) rr[i] = *reinterpret_cast<const bf16x4*>(dres + row * d + c);
                if (nxt < rows) {
                    dyn[i] = *reinterpret_cast<const bf16x4*>(dy + nxt * d + c);
                    xn[i] = *reinterpret_cast<const bf16x4*>(xin + nxt * d + c);
                }
            }
        }
        if (nxt < rows) {
            mun = mean[nxt];
            rsn = rstd[nxt];
        }
        f32x4 xh[NCH], gy[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                const f32x4 dyv = cvt(dyr[i]);
                xh[i] = (cvt(xr[i]) - mu) * rs;
                gy[i] = dyv * load4(gamma + c);
                dg[i] += dyv * xh[i];
                db[i] += dyv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 += gy[i][j];
                    s2 += gy[i][j] * xh[i][j];
                }
            }
        }
        s1 = wave_sum(s1) / d;
        s2 = wave_sum(s2) / d;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (c < d) {
                f32x4 o = (gy[i] - s1 - xh[i] * s2) * rs;
                if (dres) o += cvt(rr[i]);
                store4(dx + row * d + c, o);
                if constexpr (EX) {
                    f32x4 od = o;
                    if (dx_drop) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) od[j] = to_f32(from_f32<bf16_t>(o[j])) * drop_mul(drop, (uint32_t)(row * d + c + j));
                        store4(dx_drop + row * d + c, od);
                    }
                    if (q8.out) {
                        float r[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            r[j] = to_f32(from_f32<bf16_t>(od[j]));
                            q8m = fmaxf(q8m, fabsf(r[j]));
                        }
                        *reinterpret_cast<int*>(q8.out + row * d + c) = q8_pack4(q8, q8sc, r[0], r[1], r[2], r[3]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            dyr[i] = dyn[i];
            xr[i] = xn[i];
        }
        mu = mun;
        rs = rsn;
        row = nxt;
    }
    if constexpr (EX) {
        if (q8.out) q8_amax(q8, q8m);
    }
    for (int pass = 0; pass < 2; ++pass) {
        for (int w = 0; w < 4; ++w) {
            __syncthreads();
            if (wave == w) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = (lane + 64 * i) * 4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float val = pass == 0 ? dg[i][j] : db[i][j];
                        red[c + j] = w == 0 ? val : red[c + j] + val;
                    }
                }
            }
        }
        __syncthreads();
        float* out = pass == 0 ? dgamma : dbeta;
        for (int c = threadIdx.x; c < d; c += 256) atomicAdd(out + c, red[c]);
    }
}

// ---- bf16 rows with 16-byte accesses.  8 bf16 per lane and access instead of 4: 8-byte accesses stream at 0.54-0.70x the
// 16-byte rate on this chip (the 4-element kernels above measure 3.9-4.0 TB/s on [147712, 768]).  A row is C = d / 8 chunks;
// a wave owns R consecutive rows such that R C = 64 NS: lane l of slot s holds chunk (64 s + l) % C of row (64 s + l) / C
// (d = 768: two rows in three slots; d = 512 / 1024: one row in one / two slots; d = 256: two rows in one slot).
template <int NS, int R>
struct Rows16 {
    int col[NS];      // first column of this lane's chunk in slot s
    int rsel[NS];     // which of the wave's R rows it belongs to
    __device__ __forceinline__ void init(int lane, int C) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int flat = 64 * s + lane;
            rsel[s] = R == 1 ? 0 : (flat >= C ? 1 : 0);
            col[s] = (flat - rsel[s] * C) * 8;
        }
    }
};

__device__ __forceinline__ void load8f(const bf16_t* p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
__device__ __forceinline__ void store8f(bf16_t* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16_t)v[j];
    *reinterpret_cast<bf16x8*>(p) = t;
}

template <int NS, int R, bool EX>
__global__ __launch_bounds__(256) void layernorm_fwd16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ add,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              bf16_t* __restrict__ y, bf16_t* __restrict__ sum_out,
                                                              float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                              int d, float eps, DropCfg drop, Q8Out q8) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    Rows16<NS, R> m;
    m.init(lane, d >> 3);
    float v[NS][8];
    float ps[R];
#pragma unroll
    for (int r = 0; r < R; ++r) ps[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int64_t row = row0 + m.rsel[s];
        float t = 0.f;
        if (row < rows) {
            const int64_t o = row * d + m.col[s];
            load8f(x + o, v[s]);
            if (EX && drop.thr) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[s][j] = (float)(bf16_t)(v[s][j] * drop_mul(drop, (uint32_t)(o + j)));
            }
            if (add) {
                float a[8];
                load8f(add + o, a);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[s][j] += a[j];
            }
            if (sum_out) {           // normalise exactly what is stored (and later handed to the backward as x_in)
                store8f(sum_out + o, v[s]);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[s][j] = (float)(bf16_t)v[s][j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) t += v[s][j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[s][j] = 0.f;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ps[r] += m.rsel[s] == r ? t : 0.f;
    }
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) mu[r] = wave_sum(ps[r]) / d;
#pragma unroll
    for (int r = 0; r < R; ++r) ps[r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float mm = R == 1 ? mu[0] : (m.rsel[s] ? mu[R - 1] : mu[0]);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float c = v[s][j] - mm;
            t += c * c;
        }
        if (row0 + m.rsel[s] >= rows) t = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) ps[r] += m.rsel[s] == r ? t : 0.f;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) rs[r] = rsqrtf(wave_sum(ps[r]) / d + eps);
    if (lane < R && row0 + lane < rows) {
        mean[row0 + lane] = lane == 0 ? mu[0] : mu[R - 1];
        rstd[row0 + lane] = lane == 0 ? rs[0] : rs[R - 1];
    }
    [[maybe_unused]] float q8sc = 1.f, q8m = 0.f;
    if constexpr (EX) q8sc = q8.out ? q8.site[1] : 1.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int64_t row = row0 + m.rsel[s];
        if (row >= rows) continue;
        const float mm = R == 1 ? mu[0] : (m.rsel[s] ? mu[R - 1] : mu[0]), rr = R == 1 ? rs[0] : (m.rsel[s] ? rs[R - 1] : rs[0]);
        const f32x4 g0 = load4(gamma + m.col[s]), g1 = load4(gamma + m.col[s] + 4);
        const f32x4 b0 = load4(beta + m.col[s]), b1 = load4(beta + m.col[s] + 4);
        float o[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = (v[s][j] - mm) * rr * g0[j] + b0[j];
            o[4 + j] = (v[s][4 + j] - mm) * rr * g1[j] + b1[j];
        }
        if (y) store8f(y + row * d + m.col[s], o);          // (null: only the fp8 copy is wanted - leccr_layernorm_fwd_ex)
        if (EX && q8.out) {
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                r[j] = (float)(bf16_t)o[j];
                q8m = fmaxf(q8m, fabsf(r[j]));
            }
            *reinterpret_cast<int2*>(q8.out + row * d + m.col[s]) =
                int2{q8_pack4(q8, q8sc, r[0], r[1], r[2], r[3]), q8_pack4(q8, q8sc, r[4], r[5], r[6], r[7])};
        }
    }
    if (EX && q8.out) q8_amax(q8, q8m);
}

// (R, NS) of the 16-byte forward kernel for a row of d bf16, or false: the 4-element kernel takes the shape.  (A backward with
// the same mapping needs 150-216 registers for its three sets of column partials and measured slower than the 4-element one.)
static bool rows16_shape(int64_t d, int* R, int* NS) {
    if (d & 7) return false;
    const int64_t C = d >> 3;
    if (C % 64 == 0 && C / 64 <= 4) { *R = 1; *NS = (int)(C / 64); return true; }
    if ((2 * C) % 64 == 0 && C < 64 * 3 && 2 * C / 64 <= 5) { *R = 2; *NS = (int)(2 * C / 64); return true; }
    return false;
}

template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         float* __restrict__ inv_norm, int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float v = to_f32(x[row * d + c]);
        s += v * v;
    }
    const float inv = 1.f / fmaxf(sqrtf(wave_sum(s)), 1e-12f);  // F.normalize eps
    if (lane == 0) inv_norm[row] = inv;
    for (int c = lane; c < d; c += 64) y[row * d + c] = from_f32<T>(to_f32(x[row * d + c]) * inv);
}

// y = x * inv  ->  dx = inv * (dy - y * <dy, y>)
template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                         const float* __restrict__ inv_norm, T* __restrict__ dx,
                                                         int64_t rows, int d) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) s += to_f32(dy[row * d + c]) * to_f32(y[row * d + c]);
    s = wave_sum(s);
    const float inv = inv_norm[row];
    for (int c = lane; c < d; c += 64)
        dx[row * d + c] = from_f32<T>(inv * (to_f32(dy[row * d + c]) - to_f32(y[row * d + c]) * s));
}

// out[n] += sum_m x[m, n].  Block = 64 column-quads x 4 row phases; grid.y slices the rows.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, float* __restrict__ out, int64_t M,
                                                     int64_t N, int64_t ldx, int rows_per_block) {
    __shared__ float red[4][256 + 4];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t n = ((int64_t)blockIdx.x * 64 + tx) * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (n + 3 < N) {
        for (int64_t m = r0 + ty; m < r1; m += 4) acc += load4(x + m * ldx + n);
    } else {
        for (int64_t m = r0 + ty; m < r1; m += 4)
            for (int j = 0; j < 4; ++j)
                if (n + j < N) acc[j] += to_f32(x[m * ldx + n + j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[ty][tx * 4 + j] = acc[j];
    __syncthreads();
    const int c = threadIdx.x;
    const int64_t nn = (int64_t)blockIdx.x * 256 + c;
    if (nn < N) atomicAdd(out + nn, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

// bf16 with 16-byte accesses: block = 32 column-octets x 8 row phases (N % 8 == 0, ldx % 8 == 0)
__global__ __launch_bounds__(256) void colsum16_kernel(const bf16_t* __restrict__ x, float* __restrict__ out, int64_t M, int64_t N,
                                                       int64_t ldx, int rows_per_block) {
    __shared__ float red[8][256 + 8];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t n = ((int64_t)blockIdx.x * 32 + tx) * 8;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (n < N) {
        const bf16_t* p = x + n;
        int64_t m = r0 + ty;
        for (; m + 8 < r1; m += 16) {                 // two independent loads per iteration
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(p + m * ldx), b = *reinterpret_cast<const bf16x8*>(p + (m + 8) * ldx);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)a[j] + (float)b[j];
        }
        for (; m < r1; m += 8) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(p + m * ldx);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)a[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[ty][tx * 8 + j] = acc[j];
    __syncthreads();
    const int c = threadIdx.x;
    const int64_t nn = (int64_t)blockIdx.x * 256 + c;
    if (nn < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][c];
        atomicAdd(out + nn, t);
    }
}

template <typename T>
static int ln_fwd_dispatch(const void* x, const void* add, const float* gamma, const float* beta, void* y,
                           void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps,
                           const DropCfg& drop, const Q8Out& q8, hipStream_t st) {
    const bool ex = drop.thr != 0 || q8.out != nullptr;
    if constexpr (sizeof(T) == 2) {
        int R = 0, NS = 0;
        if (rows16_shape(d, &R, &NS)) {
            dim3 g16((unsigned)cdiv(cdiv(rows, R), 4));
#define F16E(NS_, R_, EX_)                                                                                                    \
    hipLaunchKernelGGL((layernorm_fwd16_kernel<NS_, R_, EX_>), g16, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)add, gamma,  \
                       beta, (bf16_t*)y, (bf16_t*)sum_out, mean, rstd, rows, (int)d, eps, drop, q8)
#define F16(NS_, R_)                 \
    do {                             \
        if (ex) F16E(NS_, R_, true); \
        else F16E(NS_, R_, false);   \
    } while (0)
            if (R == 1 && NS == 1) F16(1, 1); else if (R == 1 && NS == 2) F16(2, 1); else if (R == 1 && NS == 3) F16(3, 1);
            else if (R == 1 && NS == 4) F16(4, 1); else if (R == 2 && NS == 1) F16(1, 2); else if (R == 2 && NS == 3) F16(3, 2);
            else if (R == 2 && NS == 5) F16(5, 2); else R = 0;
#undef F16
#undef F16E
            if (R) return check_launch("layernorm_fwd");
        }
    }
    dim3 grid((unsigned)cdiv(rows, 4));
#define LE(N, EX_)                                                                                                      \
    hipLaunchKernelGGL((layernorm_fwd_kernel<T, N, EX_>), grid, dim3(256), 0, st, (const T*)x, (const T*)add, gamma, beta, \
                       (T*)y, (T*)sum_out, mean, rstd, rows, (int)d, eps, drop, q8)
#define L(N)                  \
    do {                      \
        if (ex) LE(N, true);  \
        else LE(N, false);    \
    } while (0)
    if (d <= 256) L(1); else if (d <= 512) L(2); else if (d <= 768) L(3); else if (d <= 1024) L(4);
    else if (d <= 2048) L(8); else L(16);
#undef L
#undef LE
    return check_launch("layernorm_fwd");
}

template <typename T>
static int ln_bwd_dispatch(const void* dy, const void* xin, const float* gamma, const float* mean, const float* rstd,
                           const void* dres, void* dx, float* dgamma, float* dbeta, float* colsum, int colsum_of,
                           int64_t rows, int64_t d, void* dx_drop, const DropCfg& drop, const Q8Out& q8, hipStream_t st) {
    int64_t nb = cdiv(rows, 4);
    const bool ex = dx_drop != nullptr || q8.out != nullptr;
    if constexpr (std::is_same<T, bf16_t>::value) {
        // the pipelined kernel: plain bf16 launches of the 768-wide (and narrower) rows with enough rows to pipeline over (LECCR_LN_BWD_PIPE=0: the plain kernel)
        static const int pipe = [] { const char* e = getenv("LECCR_LN_BWD_PIPE"); return e ? atoi(e) : 1; }();
        // (round 5, second step: the EX outputs and the 1024-wide rows too - 3 workgroups per CU there; LECCR_LN_BWD_PIPE=2: only the first step's launches)
        if (pipe && !colsum_of && d <= 1024 && rows >= 8192 && (pipe != 2 || (!ex && d <= 768))) {
            const int64_t cap = d <= 768 ? 1024 : 768;      // workgroups per CU x 256: 4 up to 768-wide rows (116 - 128 registers), 3 at 1024 (142 - 152)
            const int64_t g = nb > cap ? cap : nb;
#define LP(N, EX_) hipLaunchKernelGGL((layernorm_bwd_pipe_kernel<N, EX_>), dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)xin, gamma, mean, rstd, \
                                 (const bf16_t*)dres, (bf16_t*)dx, dgamma, dbeta, rows, (int)d, (bf16_t*)dx_drop, drop, q8)
            if (ex) { if (d <= 256) LP(1, true); else if (d <= 512) LP(2, true); else if (d <= 768) LP(3, true); else LP(4, true); }
            else { if (d <= 256) LP(1, false); else if (d <= 512) LP(2, false); else if (d <= 768) LP(3, false); else LP(4, false); }
#undef LP
            return check_launch("layernorm_bwd (pipelined)");
        }
    }
    // 4 / 5 workgroups per CU fit (registers): exactly one round.  768 < d <= 1024 (four chunks per lane: 110 registers) fits 4 - with 1280
    // workgroups a fifth of them ran as a second round: 456 us against 356 us for the (4-per-CU) EX instantiation at [186880, 1024]
    const int64_t resident = (colsum_of || ex || (d > 768 && d <= 1024)) ? 1024 : 1280;
    if (nb > resident) nb = resident;
    dim3 grid((unsigned)nb);
#define LB(N, CS_, EX_)                                                                                                         \
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, N, CS_, EX_>), grid, dim3(256), 0, st, (const T*)dy, (const T*)xin, gamma, mean, rstd, \
                       (const T*)dres, (T*)dx, dgamma, dbeta, colsum, colsum_of, rows, (int)d, (T*)dx_drop, drop, q8)
#define L(N)                                     \
    do {                                         \
        if (colsum_of) LB(N, true, false);       \
        else if (ex) LB(N, false, true);         \
        else LB(N, false, false);                \
    } while (0)
    if (d <= 256) L(1); else if (d <= 512) L(2); else if (d <= 768) L(3); else if (d <= 1024) L(4);
    else if (d <= 2048) L(8); else L(16);
#undef L
#undef LB
    return check_launch("layernorm_bwd");
}

}  // namespace leccr

using namespace leccr;

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)                           \
    do {                                                                 \
        if ((dtype) == LECCR_F32) return CALL_F32;                       \
        if ((dtype) == LECCR_BF16) return CALL_BF16;                     \
        set_error("bad dtype %d", (int)(dtype));                         \
        return LECCR_EINVAL;                                             \
    } while (0)

static int q8_check(const char* who, void* q8, float* site, int fmt, int64_t d) {
    if (!q8) return LECCR_OK;
    LECCR_REQUIRE(site && (fmt == 0 || fmt == 1), LECCR_EINVAL, "%s: the fp8 copy needs its site and fmt 0 (e4m3) / 1 (e5m2)", who);
    LECCR_REQUIRE((d & 7) == 0 && ((uintptr_t)q8 & 7) == 0, LECCR_EALIGN, "%s: the fp8 copy needs d %% 8 == 0 and an 8-byte aligned buffer", who);
    return LECCR_OK;
}

static int ln_fwd_entry(const void* x, const void* add, const float* gamma, const float* beta, void* y, void* sum_out,
                        float* mean, float* rstd, int64_t rows, int64_t d, float eps, float p, uint64_t seed, void* y8,
                        float* y8_site, int y8_fmt, int dtype, void* stream) {
    if (y8_fmt >= 2) {          // the bf16 split copy (leccr_layernorm_fwd_s3): f32 rows, 8-byte aligned copy, no site; y itself may be omitted
        LECCR_REQUIRE(y8 && (y8_fmt == 2 || y8_fmt == 3) && dtype == LECCR_F32 && ((uintptr_t)y8 & 7) == 0, LECCR_EINVAL,
                      "layernorm_fwd_s3: needs the copy's buffer (8-byte aligned), form 0 / 1, f32 rows");
    } else {
        if (int rc = q8_check("layernorm_fwd", y8, y8_site, y8_fmt, d)) return rc;
        LECCR_REQUIRE(y || y8, LECCR_EINVAL, "layernorm_fwd: null y (it may be omitted only beside its fp8 copy)");
    }
    const Q8Out q8{(uint8_t*)y8, y8_site, y8_fmt};
    LECCR_REQUIRE(x && gamma && beta && mean && rstd, LECCR_EINVAL, "layernorm_fwd: null pointer");
    LECCR_REQUIRE(d > 0 && d <= 4096 && (d & 3) == 0, LECCR_EINVAL, "layernorm_fwd: d=%lld must be a multiple of 4, <= 4096", (long long)d);
    LECCR_REQUIRE(p >= 0.f && p < 1.f && (p == 0.f || rows * d < (1ll << 32)), LECCR_EINVAL,
                  "layernorm_fwd: dropout p in [0, 1); with dropout fewer than 2^32 elements (32-bit element counter of the mask)");
    if (rows == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    DropCfg drop = make_drop(p, seed);
    if (p == 0.f) drop.thr = 0;
    const double esz = dtype == LECCR_BF16 ? 2.0 : 4.0;
    ProfScope prof(PROF_LN_FWD, (double)rows * d * (esz * (1 + (y ? 1 : 0) + (add ? 1 : 0) + (sum_out ? 1 : 0)) + (y8 ? (y8_fmt >= 2 ? 6.0 : 1.0) : 0.0)) + 8.0 * rows, st);
    DISPATCH_T(dtype, ln_fwd_dispatch<float>(x, add, gamma, beta, y, sum_out, mean, rstd, rows, d, eps, drop, q8, st),
               ln_fwd_dispatch<bf16_t>(x, add, gamma, beta, y, sum_out, mean, rstd, rows, d, eps, drop, q8, st));
}

extern "C" int leccr_layernorm_fwd(const void* x, const void* add, const float* gamma, const float* beta, void* y,
                                   void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps,
                                   int dtype, void* stream) {
    return ln_fwd_entry(x, add, gamma, beta, y, sum_out, mean, rstd, rows, d, eps, 0.f, 0, nullptr, nullptr, 0, dtype, stream);
}

extern "C" int leccr_layernorm_fwd_drop(const void* x, const void* add, const float* gamma, const float* beta, void* y,
                                        void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps, float p,
                                        uint64_t seed, int dtype, void* stream) {
    return ln_fwd_entry(x, add, gamma, beta, y, sum_out, mean, rstd, rows, d, eps, p, seed, nullptr, nullptr, 0, dtype, stream);
}

extern "C" int leccr_layernorm_fwd_ex(const void* x, const void* add, const float* gamma, const float* beta, void* y,
                                      void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps, float p,
                                      uint64_t seed, void* y8, float* y8_site, int y8_fmt, int dtype, void* stream) {
    LECCR_REQUIRE(!y8 || y8_fmt == 0 || y8_fmt == 1, LECCR_EINVAL, "layernorm_fwd_ex: fmt 0 (e4m3) / 1 (e5m2)");
    return ln_fwd_entry(x, add, gamma, beta, y, sum_out, mean, rstd, rows, d, eps, p, seed, y8, y8_site, y8_fmt, dtype, stream);
}

// .. and, for f32 rows, the bf16 (hi, lo) split copy of y (leccr_split3's [rows, 3 d] layout, form 0 / 1) written by the pass itself - the
// operand copy the fp32 path's next product reads; y may be NULL when nothing reads the f32 tensor (a pre-LN block's h1 / h2)
extern "C" int leccr_layernorm_fwd_s3(const void* x, const void* add, const float* gamma, const float* beta, void* y, void* sum_out,
                                      float* mean, float* rstd, int64_t rows, int64_t d, float eps, float p, uint64_t seed, void* y3,
                                      int form, void* stream) {
    LECCR_REQUIRE(form == 0 || form == 1, LECCR_EINVAL, "layernorm_fwd_s3: form 0 (hi|hi|lo) or 1 (hi|lo|hi)");
    return ln_fwd_entry(x, add, gamma, beta, y, sum_out, mean, rstd, rows, d, eps, p, seed, y3, nullptr, 2 + form, LECCR_F32, stream);
}

static int ln_bwd_entry(const void* dy, const void* x_in, const float* gamma, const float* mean, const float* rstd,
                        const void* dres, void* dx, float* dgamma, float* dbeta, float* colsum, int colsum_of, void* dx_drop,
                        float p, uint64_t seed, void* g8, float* g8_site, int g8_fmt, int64_t rows, int64_t d, int dtype,
                        void* stream) {
    if (g8 && g8_fmt >= 2) {       // the bf16 split copy of the fp32 path (leccr_layernorm_bwd_ex, fmt 2 / 3): f32 rows, 8-byte aligned copy, no site
        LECCR_REQUIRE((g8_fmt == 2 || g8_fmt == 3) && dtype == LECCR_F32 && ((uintptr_t)g8 & 7) == 0, LECCR_EINVAL,
                      "layernorm_bwd_ex: the split copy needs fmt 2 (hi|hi|lo) / 3 (hi|lo|hi), f32 rows and an 8-byte aligned buffer");
    } else if (int rc = q8_check("layernorm_bwd", g8, g8_site, g8_fmt, d)) return rc;
    const Q8Out q8{(uint8_t*)g8, g8_site, g8_fmt};
    LECCR_REQUIRE(dy && x_in && gamma && mean && rstd && dx && dgamma && dbeta, LECCR_EINVAL, "layernorm_bwd: null pointer");
    LECCR_REQUIRE(colsum_of == 0 || (colsum && (colsum_of == 2 || (colsum_of == 1 && dres))), LECCR_EINVAL,
                  "layernorm_bwd: colsum_of=%d needs colsum (and dres for 1)", colsum_of);
    LECCR_REQUIRE(d > 0 && d <= 4096 && (d & 3) == 0, LECCR_EINVAL, "layernorm_bwd: d=%lld must be a multiple of 4, <= 4096", (long long)d);
    LECCR_REQUIRE(!dx_drop || (p >= 0.f && p < 1.f && rows * d < (1ll << 32)), LECCR_EINVAL,
                  "layernorm_bwd: dropout p in [0, 1), fewer than 2^32 elements");
    if (rows == 0) return LECCR_OK;
    hipStream_t st = as_stream(stream);
    const DropCfg drop = make_drop(dx_drop ? p : 0.f, seed);
    const double esz = dtype == LECCR_BF16 ? 2.0 : 4.0;
    ProfScope prof(PROF_LN_BWD, (double)rows * d * (esz * (3 + (dres ? 1 : 0) + (dx_drop ? 1 : 0)) + (g8 ? (g8_fmt >= 2 ? 6.0 : 1.0) : 0.0)) + 8.0 * rows, st);
    DISPATCH_T(dtype, ln_bwd_dispatch<float>(dy, x_in, gamma, mean, rstd, dres, dx, dgamma, dbeta, colsum, colsum_of, rows, d, dx_drop, drop, q8, st),
               ln_bwd_dispatch<bf16_t>(dy, x_in, gamma, mean, rstd, dres, dx, dgamma, dbeta, colsum, colsum_of, rows, d, dx_drop, drop, q8, st));
}

extern "C" int leccr_layernorm_bwd(const void* dy, const void* x_in, const float* gamma, const float* mean,
                                   const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                   float* colsum, int colsum_of, int64_t rows, int64_t d, int dtype, void* stream) {
    return ln_bwd_entry(dy, x_in, gamma, mean, rstd, dres, dx, dgamma, dbeta, colsum, colsum_of, nullptr, 0.f, 0, nullptr, nullptr, 0, rows, d, dtype, stream);
}

extern "C" int leccr_layernorm_bwd_drop(const void* dy, const void* x_in, const float* gamma, const float* mean,
                                        const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                        void* dx_drop, float p, uint64_t seed, int64_t rows, int64_t d, int dtype,
                                        void* stream) {
    LECCR_REQUIRE(dx_drop, LECCR_EINVAL, "layernorm_bwd_drop: dx_drop is null");
    return ln_bwd_entry(dy, x_in, gamma, mean, rstd, dres, dx, dgamma, dbeta, nullptr, 0, dx_drop, p, seed, nullptr, nullptr, 0, rows, d, dtype, stream);
}

extern "C" int leccr_layernorm_bwd_ex(const void* dy, const void* x_in, const float* gamma, const float* mean,
                                      const float* rstd, const void* dres, void* dx, float* dgamma, float* dbeta,
                                      void* dx_drop, float p, uint64_t seed, void* g8, float* g8_site, int g8_fmt,
                                      int64_t rows, int64_t d, int dtype, void* stream) {
    return ln_bwd_entry(dy, x_in, gamma, mean, rstd, dres, dx, dgamma, dbeta, nullptr, 0, dx_drop, p, seed, g8, g8_site, g8_fmt, rows, d,
                        dtype, stream);
}

template <typename T>
static int l2f(const void* x, void* y, float* inv, int64_t rows, int64_t d, hipStream_t st) {
    hipLaunchKernelGGL((l2norm_fwd_kernel<T>), dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, (const T*)x, (T*)y, inv, rows, (int)d);
    return check_launch("l2norm_fwd");
}
template <typename T>
static int l2b(const void* dy, const void* y, const float* inv, void* dx, int64_t rows, int64_t d, hipStream_t st) {
    hipLaunchKernelGGL((l2norm_bwd_kernel<T>), dim3((unsigned)cdiv(rows, 4)), dim3(256), 0, st, (const T*)dy, (const T*)y, inv, (T*)dx, rows, (int)d);
    return check_launch("l2norm_bwd");
}

extern "C" int leccr_l2norm_fwd(const void* x, void* y, float* inv_norm, int64_t rows, int64_t d, int dtype, void* stream) {
    LECCR_REQUIRE(x && y && inv_norm && d > 0, LECCR_EINVAL, "l2norm_fwd: bad argument");
    if (rows == 0) return LECCR_OK;
    DISPATCH_T(dtype, l2f<float>(x, y, inv_norm, rows, d, as_stream(stream)), l2f<bf16_t>(x, y, inv_norm, rows, d, as_stream(stream)));
}

extern "C" int leccr_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int64_t rows, int64_t d,
                                int dtype, void* stream) {
    LECCR_REQUIRE(dy && y && inv_norm && dx && d > 0, LECCR_EINVAL, "l2norm_bwd: bad argument");
    if (rows == 0) return LECCR_OK;
    DISPATCH_T(dtype, l2b<float>(dy, y, inv_norm, dx, rows, d, as_stream(stream)), l2b<bf16_t>(dy, y, inv_norm, dx, rows, d, as_stream(stream)));
}

template <typename T>
static int cs(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, hipStream_t st) {
    int rpb = 256;
    while ((int64_t)cdiv(M, rpb) * cdiv(N, 256) > 4096) rpb *= 2;
    dim3 grid((unsigned)cdiv(N, 256), (unsigned)cdiv(M, rpb));
    if constexpr (sizeof(T) == 2) {
        if ((N & 7) == 0 && (ldx & 7) == 0) {
            hipLaunchKernelGGL(colsum16_kernel, grid, dim3(256), 0, st, (const bf16_t*)x, out, M, N, ldx, rpb);
            return check_launch("colsum");
        }
    }
    hipLaunchKernelGGL((colsum_kernel<T>), grid, dim3(256), 0, st, (const T*)x, out, M, N, ldx, rpb);
    return check_launch("colsum");
}

extern "C" int leccr_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int dtype, void* stream) {
    LECCR_REQUIRE(x && out && ldx >= N, LECCR_EINVAL, "colsum: bad argument");
    LECCR_REQUIRE((ldx & 3) == 0 && ((uintptr_t)x & 15) == 0, LECCR_EALIGN, "colsum: x must be 16-byte aligned with ldx %% 4 == 0");
    if (M == 0 || N == 0) return LECCR_OK;
    ProfScope prof(PROF_COLSUM, (double)M * N * (dtype == LECCR_BF16 ? 2.0 : 4.0), as_stream(stream));
    DISPATCH_T(dtype, cs<float>(x, out, M, N, ldx, as_stream(stream)), cs<bf16_t>(x, out, M, N, ldx, as_stream(stream)));
}
