/* leccr_hip.h - C ABI of libleccr_hip.so, the MI355X (gfx950) kernels behind the LECCR retrieval hot path.
 *
 * The reference (LiJiaBei-7/leccr) is pure Python: it has no FFI/operator layer, every op on its hot path is an
 * ATen op sequence.  Each entry point below replaces one such sequence; the comment in front of it cites the
 * reference lines it replaces (paths relative to /root/reference/LECCR).  INTEGRATION.md shows the ctypes stub
 * a maintainer of the reference would add to call them.
 *
 * Conventions
 *  - plain pointers to DEVICE memory + sizes; no torch types; the caller owns every buffer (kernels allocate
 *    nothing); `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous.
 *  - return 0 on success, a negative LECCR_E* code on a rejected argument, or -1000-hipError_t on a launch error.
 *  - dtype codes: LECCR_F32 = 0, LECCR_BF16 = 1.  "T" buffers below have the dtype passed as `dtype`.
 *    Statistics, losses, biases, LayerNorm gains and every parameter gradient are always f32.
 *  - all activation tensors are batch-major and row-major: [rows, features] with a leading dimension in
 *    ELEMENTS.  (The reference's LND layout for nn.MultiheadAttention is a layout choice only.)
 */
#ifndef LECCR_HIP_H
#define LECCR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LECCR_F32 0
#define LECCR_BF16 1
#define LECCR_FP8 2         /* leccr_gemm ab_dtype only: A and B are OCP e4m3 bytes (gfx950: e4m3fn, not MI300's fnuz) */
#define LECCR_FP8_BF8A 3    /* leccr_gemm ab_dtype only: A is OCP e5m2 (gradients), B is OCP e4m3 (weights) */

#define LECCR_OK 0
#define LECCR_EINVAL (-1)      /* bad shape / null pointer / unsupported combination */
#define LECCR_EALIGN (-2)      /* pointer or leading dimension not aligned as the kernel requires */

/* ---- library ---------------------------------------------------------------------------------------------- */
int leccr_version(void);                       /* ABI version, bumps on any signature change */
const char* leccr_last_error(void);            /* text of the last error on this thread */
int leccr_set_force_generic(int on);           /* 1: route every op through the simple f32-arithmetic kernels */
/* Optional timing of the GEMM / attention launches with hipEvents recorded on the launch stream (bench.py's
 * roofline leg).  read(): ms[5], work[5] (algorithmic FLOPs), launches[5] indexed
 * {gemm_mfma_nt, gemm_mfma_tn, gemm_generic, attn_fwd, attn_bwd}; synchronises on the recorded events. */
int leccr_set_gemm_variant(int v);             /* 0 auto (default), 1 force the 128x128 kernel, 2 / 3 force the 8-wave /
                                                * 4-wave 256x256 kernel where it is legal, 4 force the 4-wave TN (weight-gradient)
                                                * kernel where it is legal (everything else as auto) */
int leccr_set_gemm_max_blocks(int n);          /* workgroups (= CUs) a persistent 256x256 GEMM launch may use: 0 / 256 = all
                                                * (default); rounded down to a multiple of 8; returns the value in effect.
                                                * leccr_amd.parallel lowers it while an all-reduce overlaps the backward pass */
int leccr_get_gemm_max_blocks(void);
int leccr_set_gemm_max_blocks_for(int n, int64_t launches);   /* the same limit for the next `launches` persistent-GEMM launches
                                                * only, after which all CUs are used again (the expected duration of a collective) */
int leccr_last_gemm_kernel(void);              /* which kernel the last leccr_gemm launched: 0 generic, 1 MFMA 128x128,
                                                * 2 MFMA 256x256 (8 waves), 3 MFMA 256x256 (4 waves), 4 the fp8 4-wave kernel,
                                                * 5 the 4-wave TN weight-gradient kernel, 6 its fp8 form; tests use it */
int leccr_prof_enable(int on);
int leccr_prof_read(double* ms, double* work, int64_t* launches);
/* The first n <= 14 kinds: the five above, then {layernorm_fwd, layernorm_bwd, colsum, embedding stages, adamw, dropout} -
 * the HBM-bound launches, whose work[] entry is ALGORITHMIC BYTES (every operand read once, every result written once) - then
 * {fp8-operand NT GEMMs (forward / dX), fp8-operand weight-gradient GEMMs}: flops, kept apart from the bf16 launches of kinds 0 / 1
 * so that a mixed-precision step can be priced launch by launch (fp8 against the fp8 MFMA peak, bf16 against the bf16 one); then
 * {leccr_split3 passes}: bytes. */
int leccr_prof_read_n(int n, double* ms, double* work, int64_t* launches);
/* The launches of the kinds in kind_mask (bit k = index k above) as one busy time: the union of their [start, stop]
 * intervals (launches that overlap on two streams are not counted twice), their total work and count. */
int leccr_prof_read_union(int kind_mask, double* busy_ms, double* work, int64_t* launches);
int64_t leccr_prof_dropped(void);              /* launches since leccr_prof_enable(1) that found the record buffer (131 072
                                                * entries) full and were not timed: the read-outs undercount when > 0 */
int leccr_debug_set_stamps(void* device_buffer);   /* diagnostic builds only (-DLECCR_G256_STAMPS, scripts/build_variant.sh): where the
                                                * 256x256 GEMM leaves its in-kernel s_memtime stamps; a no-op pointer store in the
                                                * production build (no stamp instruction is compiled in) */

int leccr_debug_occupy(int blocks, int threads, int64_t microseconds, void* stream);   /* diagnostic: `blocks` workgroups that hold a CU slot for
                                                * that long and exit - a stand-in for a collective's resident workgroups on a one-GPU box */

/* ---- GEMM -------------------------------------------------------------------------------------------------
 * C[M,N] = epilogue( alpha * sum_k A(m,k) * B(n,k) )      (+ C if accumulate)
 *   trans_a = 0: A is [M,K] row-major (lda >= K)      trans_a = 1: A is [K,M] row-major (lda >= M)
 *   trans_b = 0: B is [N,K] row-major (nn.Linear weight layout)   trans_b = 1: B is [K,N] row-major
 * Replaces F.linear / torch.mm / addmm on every projection of the path:
 *   clip/model.py:175-191 (in_proj, out_proj, c_fc, c_proj), :211 (conv1 as a GEMM over patches), :342 (@ proj);
 *   HF BertModel dense layers called at models/xvlm.py:238; models/attention.py:28-31,47-52;
 *   models/xvlm.py:241-256 (vision_proj/text_proj); and their autograd backward (dX: trans_b=1, dW: trans_a=1,trans_b=1).
 * Epilogue order: v = alpha*acc (+bias[n]) ; if aux_out: aux_out[m,n] = v (act'(v) for the _D forms) ; v = act(v) or v * act'(aux[m,n]) or v * aux[m,n] ;
 *                 v += residual[m,n] ; C = v (or C += v).
 */
#define LECCR_EPI_NONE 0
#define LECCR_EPI_QUICKGELU 1     /* x * sigmoid(1.702 x)                    clip/model.py:166-168 */
#define LECCR_EPI_GELU 2          /* 0.5 x (1 + erf(x / sqrt 2))             attention.py:16-17, HF BERT 'gelu' */
#define LECCR_EPI_DQUICKGELU 3    /* v * d/du quickgelu(u), u = aux[m,n]     (backward through the activation) */
#define LECCR_EPI_DGELU 4         /* v * d/du gelu(u)                                                          */
/* The same two activations with the DERIVATIVE saved instead of the pre-activation (round 5): the forward epilogue has sigmoid / erf in
 * registers, the derivative is four more plain operations there, and the backward epilogue becomes one multiply instead of an exp2 and a
 * reciprocal per element after its K loop (dX of c_proj at config B: 870 -> ~690 us; the fp8 product of config E: 1418 -> ~1000 us). */
#define LECCR_EPI_QUICKGELU_D 5   /* C = quickgelu(v), aux_out[m,n] = d/dv quickgelu(v)  (aux_out required)    */
#define LECCR_EPI_GELU_D 6        /* C = gelu(v),      aux_out[m,n] = d/dv gelu(v)       (aux_out required)    */
#define LECCR_EPI_MULAUX 7        /* v * aux[m,n]      (backward through an activation whose derivative was saved) */

typedef struct leccr_gemm_params {
    const void* A;
    const void* B;
    void* C;
    int64_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t ab_dtype;            /* dtype of A and B.  LECCR_FP8 / LECCR_FP8_BF8A (BASELINE configs[4]: fp8 MFMA GEMM inputs): 1-byte
                                    operands quantised by leccr_quantize_fp8 with per-tensor scales sa, sb; pass alpha = 1 / (sa sb).
                                    trans_a = trans_b = 0, bf16 C, M, N >= 256, K % 128 == 0, lda / ldb % 16 == 0 only: the product runs on
                                    v_mfma_scale_f32_32x32x64_f8f6f4 (block scales 1.0) in the 4-wave 256x256 kernel, f32 accumulation, the forward /
                                    dX epilogues of the bf16 path.  lda / ldb / K count fp8 ELEMENTS (= bytes).
                                    LECCR_FP8_BF8A with trans_a = trans_b = 1: the weight-gradient product dW += alpha dY^T X on the same
                                    instruction - A = dY e5m2 [K, M], B = X e4m3 [K, N], both row-major as the forward / dX products hold
                                    them, f32 C with accumulate = 1, no epilogue / sums, M, N % 256 == 0, K % 128 == 0, K >= 2048. */
    int32_t c_dtype;             /* dtype of C, aux, aux_out, residual */
    int32_t trans_a, trans_b;
    const float* bias;           /* [N] f32 or NULL */
    int32_t epilogue;
    int32_t accumulate;          /* 1: C += v (c_dtype must be f32); split-K uses f32 atomics */
    const void* aux;             /* [M,N] pre-activation for the D* epilogues */
    int64_t ldaux;
    void* aux_out;               /* [M,N] optional pre-activation output (ld = ldc) */
    const void* residual;        /* [M,N] optional */
    int64_t ldr;
    float alpha;
    int32_t reserved;
    float* colsum;               /* optional [N] f32: += column sums of the C written (the bias gradient when C is a
                                    dY); currently one extra pass over C on the same stream */
    float* asum;                 /* optional [M] f32, trans_a = 1 only: asum[m] += alpha * sum_k A(m,k) - the column sums of
                                    the stored [K,M] matrix over the reduction index.  When the product is a weight gradient
                                    dW = dY^T X this is the bias gradient colsum(dY) (autograd of F.linear's bias: clip/model.py:
                                    175-191, HF BERT dense layers); the 256x256 kernel computes it on the matrix cores beside
                                    the product (an all-ones B fragment), other paths run leccr_colsum over A.  With fp8 operands
                                    (LECCR_FP8_BF8A, trans_a = trans_b = 1): alpha * alpha_dev[0] * the column sums of the e5m2
                                    bytes - the bias gradient built from the same quantised dY as the weight gradient beside it */
    const float* alpha_dev[2];   /* optional device scalars multiplied into alpha when the kernel starts (fp8 operands: the two
                                    inverse quantisation scales live on the device, leccr_fp8_* below; no host read) */
    void* c8;                    /* optional fp8 copy of the C written, for the NEXT product that consumes it (fp8 operand kernels
                                    with an activation / d-activation epilogue only): c8[m,n] = fp8(clamp(C[m,n] * c8_site.scale)),
                                    row stride ldc8 bytes, format c8_fmt (0 e4m3, 1 e5m2); c8_site.amax accumulates max |C| - the
                                    same contract as leccr_fp8_quantize, without its pass over C.  With c8, C itself may be NULL: a
                                    consumer chain that is fp8 end to end reads only the copy, and the bf16 C stays unwritten (2 of the
                                    epilogue's 5 bytes per element).  With c8 on the dX form (LECCR_FP8_BF8A, derivative epilogue),
                                    `colsum` is fused: colsum[n] += sum_m of the f32 epilogue values (the bias gradient of the Linear
                                    that produced the pre-activation), so nothing needs the bf16 C */
    int64_t ldc8;
    float* c8_site;
    int32_t c8_fmt;
    int32_t reserved2;
    void* c3;                    /* optional bf16 (hi, lo) split copy of the C written, [M, 3 N] in the form c3_form of leccr_split3 (0: hi|hi|lo,
                                    1: hi|lo|hi) - the operand copy the NEXT product of the fp32 path consumes, without leccr_split3's pass over
                                    C.  bf16 operands, f32 C, trans_a = trans_b = 0, no accumulate.  The 8-wave 256x256 kernel writes it from its
                                    epilogue registers; every other kernel is followed by a leccr_split3 pass over C here.  With c3, C itself
                                    may be NULL when the 256x256 kernel takes the shape (M, N >= 256, >= 192 tiles, N % 8 == 0): a consumer
                                    chain that reads only the copy (the 4d-wide MLP activation and its gradient) */
    int32_t c3_form;
    int32_t asum_x3;             /* 1 (with asum; bf16 operands, trans_a = 1, K % 3 == 0): A is the 3-segment G-form split copy [rows, 3 M] (hi | lo | hi,
                                    leccr_split3) of an f32 dY read as [3 rows, M] - the fp32 path's weight-gradient product - and asum adds its hi and lo
                                    segments ONCE each (reduction indices k with k mod 3 = 0, 1): asum[m] += alpha * sum_r (hi + lo)(r, m) = the bias
                                    gradient colsum(dY) to the split's 2^-17, inside the weight-gradient launch (the 4-wave TN kernel weights the
                                    fragment pairs of its column sums by k mod 3; other kernels: two leccr_colsum passes over the copy's first two
                                    column blocks).  Was `reserved3` (0 = the plain sum). */
} leccr_gemm_params;

int leccr_gemm(const leccr_gemm_params* p, void* stream);

/* ---- fp32 operands on the bf16 matrix cores ("bf16x3") ---------------------------------------------------------
 * The reference's arithmetic is fp32 (models/xvlm.py:88 `clip_encoder.float()`, clip/model.py:171-192; HF BertModel in fp32).  An
 * fp32 x is hi + lo + e, hi = bf16(x), lo = bf16(x - hi), |e| <= 2^-18 |x|, and x w = hi_x hi_w + hi_x lo_w + lo_x hi_w + O(2^-17 |x w|):
 * three bf16 products accumulated in f32 = ONE bf16 leccr_gemm over a reduction index three times as long.  leccr_split3 writes
 * the operand copies: dst [rows, 3 * cols] bf16, every row = three segments of `cols`,
 *   form 0 ("X": activations; the [in, out] weight copy of the dX product):  hi | hi | lo
 *   form 1 ("G": gradients; the [out, in] weight of the forward product):    hi | lo | hi
 * so that an X-form operand against a G-form operand pairs up as hh + hl + lh whichever is A or B (NT: K' = 3 K, lda = ldb = 3 K).
 * The same memory read as [3 * rows, cols] is the row-interleaved form of the weight-gradient product (TN: K' = 3 * rows).
 * transpose = 1: the split of src^T, dst [cols, 3 * rows].  src f32, row stride ld_src elements.
 * colsum (optional, transpose = 0, cols % 8 == 0, cols <= 8192): colsum[c] += sum_r src[r][c] in the same pass - the bias gradient
 * when src is a dY (the column sums of the split copy itself would count hi twice). */
int leccr_split3(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int form, int transpose, float* colsum,
                 void* stream);
/* nseg = 3: leccr_split3.  nseg = 6 (the precision option LECCR_F32_SPLITS=6): three bf16 pieces p0 + p1 + p2 of every element and the six
 * products of order >= 2^-16 (p0 q0, p0 q1, p1 q0, p0 q2, p1 q1, p2 q0: 2^-24 relative, fp32-grade) as [rows, 6 * cols] copies -
 * X form p0|p0|p1|p0|p1|p2, G form p0|p1|p0|p2|p1|p0 - run by the same bf16 leccr_gemm over 6 K (a sixth of the bf16 rate, still 2.6 x the
 * f32 MFMA instruction's peak).  Untuned (one element per thread): what it is for is to show that the few tolerances the 3-segment form
 * needed widened are the form's own noise (tests/test_model_gpu.py). */
int leccr_split_n(const float* src, void* dst, int64_t rows, int64_t cols, int64_t ld_src, int form, int nseg, int transpose, float* colsum,
                  void* stream);
/* The two split copies every weight of the fp32 path needs - [out, 3 in] in form 1 for the forward product (clip/model.py:175-191, HF BERT dense layers:
 * y = x W^T) and [in, 3 out] in form 0 of W^T for the dX product - for ALL weights in one launch, once per optimizer step (the counterpart of
 * leccr_transpose_cast_batched on the bf16 path).  table: n rows of 6 int64 on the device = {first 64 x 64 tile of this matrix in the launch, src
 * pointer (f32 [R, C] contiguous), dstG pointer (bf16 [R, 3 C]), dstT pointer (bf16 [C, 3 R]), R, C}, ordered by first tile; total_tiles = the grid. */
int leccr_split3_weights_batched(const int64_t* table, int64_t n, int64_t total_tiles, void* stream);

/* ---- LayerNorm --------------------------------------------------------------------------------------------
 * y = (x - mean) * rstd * gamma + beta over the last axis, statistics in f32.
 * Replaces clip/model.py:157-163 (eps 1e-5), HF BERT LayerNorm (eps 1e-12), attention.py:35,53 norm2.
 * fwd: x [rows, d] (T, ldx) optionally plus `add` [rows,d] (T, same ld) summed first (post-LN residual:
 *      y = LN(x + add)); writes y (T), mean/rstd [rows] f32, and, if sum_out != NULL, x+add (T).
 * bwd: dx = LN'(dy) (+ dres if non-null), dgamma/dbeta ACCUMULATED (+=) into f32 [d] buffers.
 *      x_in must be the tensor that was normalised (x+add).
 *      colsum_of != 0: additionally colsum[d] (f32) += column sums of dres (1) or of the dx written (2) - the bias
 *      gradient of the Linear whose output gradient that tensor is (pre-LN blocks: dres; post-LN blocks: dx), so
 *      the separate leccr_colsum pass over it is not needed.
 */
int leccr_layernorm_fwd(const void* x, const void* add, const float* gamma, const float* beta, void* y,
                        void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps, int dtype,
                        void* stream);
int leccr_layernorm_bwd(const void* dy, const void* x_in, const float* gamma, const float* mean, const float* rstd,
                        const void* dres, void* dx, float* dgamma, float* dbeta, float* colsum, int colsum_of,
                        int64_t rows, int64_t d, int dtype, void* stream);
/* The post-LN residual blocks of HF BERT (BertSelfOutput / BertOutput, called at xvlm.py:238) and of CrossAttentionLayer
 * (attention.py:52-54) are  y = LN(dropout(dense(h)) + residual): the _drop forms take the dropout inside the LayerNorm pass
 * instead of a separate leccr_dropout launch over the tensor on each side.
 * fwd_drop: as leccr_layernorm_fwd with x replaced by dropout(x; p, seed) - the mask and rounding of leccr_dropout(x, .., p, seed)
 *           (element i of the contiguous [rows, d] tensor; rows * d < 2^32), so the results equal the two-call sequence bit for bit.
 * bwd_drop: as leccr_layernorm_bwd (colsum_of = 0) with a second output dx_drop = dropout(dx; p, seed), the gradient the dense
 *           layer sees; dx itself (the residual branch's gradient) is written as before. */
int leccr_layernorm_fwd_drop(const void* x, const void* add, const float* gamma, const float* beta, void* y,
                             void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps, float p,
                             uint64_t seed, int dtype, void* stream);
int leccr_layernorm_bwd_drop(const void* dy, const void* x_in, const float* gamma, const float* mean, const float* rstd,
                             const void* dres, void* dx, float* dgamma, float* dbeta, void* dx_drop, float p, uint64_t seed,
                             int64_t rows, int64_t d, int dtype, void* stream);
/* _ex: the _drop forms (p = 0 / dx_drop = NULL: no dropout) plus an optional fp8 copy of the output for the scaled-MFMA product that
 * consumes it (compute_dtype 'fp8', "fp8 GEMM operands" below): y8 / g8 [rows, d] bytes (8-byte aligned, d % 8 == 0), quantised under
 * `site` = {amax, scale, 1/scale, fmt} exactly as leccr_fp8_quantize would quantise the tensor just written (fwd: y, e4m3 activations;
 * bwd: dx_drop if given else dx, e5m2 gradients), |value| folded into the site's running amax - the separate quantise pass over the
 * tensor is not needed.  NULL y8 / g8: off. 
 * leccr_layernorm_fwd_ex with y = NULL (round 4): only the fp8 copy is written (a pre-LN block's h1 / h2 in the fp8 step are read by nothing else). */
int leccr_layernorm_fwd_ex(const void* x, const void* add, const float* gamma, const float* beta, void* y,
                           void* sum_out, float* mean, float* rstd, int64_t rows, int64_t d, float eps, float p,
                           uint64_t seed, void* y8, float* y8_site, int y8_fmt, int dtype, void* stream);
/* fp32 path: the pass also writes y's bf16 (hi, lo) split copy y3 [rows, 3 d] (leccr_split3's layout, form 0 / 1) - the operand copy of the
 * next product; y itself may be NULL when nothing else reads it.  f32 rows only. */
int leccr_layernorm_fwd_s3(const void* x, const void* add, const float* gamma, const float* beta, void* y, void* sum_out, float* mean,
                           float* rstd, int64_t rows, int64_t d, float eps, float p, uint64_t seed, void* y3, int form, void* stream);
/* leccr_layernorm_bwd_ex with g8_fmt 2 / 3 and g8_site = NULL (round 5, f32 rows only): g8 is the bf16 (hi, lo) split copy [rows, 3 d] of the gradient
 * the dense layer in front of the residual add sees (dx_drop if given, else dx) in leccr_split3's form g8_fmt - 2 (0: hi|hi|lo, 1: hi|lo|hi) - the
 * operand copy of that Linear's dX and weight-gradient products on the fp32 path, without leccr_split3's pass over the tensor just written
 * (the bias gradient then rides in the weight-gradient launch: leccr_gemm_params.asum_x3). */
int leccr_layernorm_bwd_ex(const void* dy, const void* x_in, const float* gamma, const float* mean, const float* rstd,
                           const void* dres, void* dx, float* dgamma, float* dbeta, void* dx_drop, float p, uint64_t seed,
                           void* g8, float* g8_site, int g8_fmt, int64_t rows, int64_t d, int dtype, void* stream);

/* ---- Attention --------------------------------------------------------------------------------------------
 * softmax(Q K^T / sqrt(hd) + key mask) V per (batch, head); no materialised score matrix.
 * Replaces nn.MultiheadAttention's core (clip/model.py:187; attention.py:43-46,103-106) and HF BertSelfAttention
 * (xvlm.py:238).  Q [b, Lq, heads*hd] with row stride ldq (elements) and batch stride Lq*ldq; K,V likewise
 * with Lk (so packed qkv projections are addressed in place).  kmask: [b, Lk] uint8, 1 = ignore key, or NULL.
 * lse [b, heads, Lq] f32 = log-sum-exp of the scaled scores (saved for backward).
 * A row whose keys are all masked returns zeros (the reference would return NaN; never happens on the path:
 * [CLS] is always unmasked).
 * Dropout on the attention probabilities (nn.MultiheadAttention(dropout=0.1) at attention.py:27,90; HF
 * attention_probs_dropout_prob): drop_p in [0,1), keep(b,h,q,k) = hash(seed, ((b*heads+h)*Lq+q)*Lk+k) >= drop_p*2^32,
 * kept probabilities scaled by 1/(1-drop_p) AFTER the softmax normalisation (the row sum uses the undropped values);
 * the backward regenerates the mask from the same seed.  drop_p = 0 disables it.
 */
int leccr_last_attn_kernel(void);              /* which family the last leccr_attn_fwd / _bwd launched: 0 generic (f32, other shapes),
                                                * 1 MFMA flash kernels (bf16, head 64), 2 thin-K, 3 thin-Q (bf16, <= 8 keys / queries per
                                                * batch element at d = heads * hd in {512, 768, 1024}: attention_thin.hip), 4 f32 flash kernels
                                                * on v_mfma_f32_32x32x2_f32 (f32, head 64, LECCR_ATTN_F32=exact: attention_f32.hip), 5 f32 flash
                                                * kernels with every operand as a bf16 (hi, lo) pair on v_mfma_f32_32x32x16_bf16 (f32, head 64,
                                                * default: attention_x3.hip); tests use it */
int leccr_set_attn_variant(int v);            /* bf16 head-64 MFMA kernels: 32 = 32 rows per wave, v_mfma_f32_32x32x16_bf16 (default),
                                                * 16 = the 16-row kernels (v_mfma_f32_16x16x32_bf16); 0 = back to the default
                                                * (environment LECCR_ATTN_FWD); returns the variant in effect.  Tests / benchmarks */
int leccr_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask,
                   int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk,
                   int64_t ldv, int64_t ldo, int dtype, float drop_p, uint64_t seed, void* stream);
/* dq/dk/dv have the same strides as q/k/v.  delta is f32 workspace of 2 * b * heads * roundup(Lq, 64) elements
 * (per-query row constants of the backward: -rowsum(dO*O) and -lse/scale, padded so tiles load without clamping). */
int leccr_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                   const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, int64_t b, int64_t heads,
                   int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                   int dtype, float drop_p, uint64_t seed, void* stream);
/* _causal: self-attention in which query i sees keys j <= i only, no dropout - CLIP's text transformer (clip/model.py:328-334, 354),
 * the caption encoder of caption_encoder_name = 'clip' (models/model_retrieval_caption.py:57-58).  Generic kernels (a 77-token problem). */
int leccr_attn_fwd_causal(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, int64_t b,
                          int64_t heads, int64_t L, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int dtype,
                          void* stream);
int leccr_attn_bwd_causal(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                          const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, int64_t b, int64_t heads, int64_t L,
                          int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, int dtype, void* stream);
/* _q8 (compute_dtype 'fp8'): the same calls plus per-tensor scaled fp8 copies of the outputs for the scaled-MFMA products that consume
 * them, under `site` = {amax, scale, 1/scale, fmt} with the contract of leccr_fp8_quantize (quantised from the f32 values, site.amax
 * accumulates max |value|): o8 [b*Lq, heads*hd] e4m3 (the out_proj input), dq8 / dk8 / dv8 e5m2 under ONE site (the three column
 * blocks of the packed in_proj gradient; row stride ld8 bytes).  Written by the attention kernels' own epilogues where the variant that
 * runs has them (MFMA head-64 kernels: the 32-query forward kernels, every backward), by quantise passes otherwise.
 * leccr_attn_bwd_q8 with dq = dk = dv = NULL (round 4): only the e5m2 copies are written - for the fp8 step, where the in_proj dX and
 * weight-gradient products (and the bias sums inside the latter) read nothing else and the bf16 gradient is 2/3 of the kernels' output
 * bytes; accepted where the head-64 bf16 MFMA kernels run the problem (else LECCR_EINVAL). */
int leccr_attn_fwd_q8(const void* q, const void* k, const void* v, void* o, float* lse, const uint8_t* kmask, int64_t b,
                      int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                      int dtype, float drop_p, uint64_t seed, void* o8, int64_t ldo8, float* site, void* stream);
int leccr_attn_bwd_q8(const void* q, const void* k, const void* v, const void* o, const void* dout, const float* lse,
                      const uint8_t* kmask, void* dq, void* dk, void* dv, float* delta, int64_t b, int64_t heads,
                      int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                      int dtype, float drop_p, uint64_t seed, void* dq8, void* dk8, void* dv8, int64_t ld8, float* site, void* stream);

/* fp32 attention that writes the OPERAND COPIES of its outputs (round 4; leccr_amd/csrc/attention.hip, attention_x3.hip).  In the fp32 path
 * (operands as bf16 hi / lo pairs, leccr_split3) o is read only as the X-form copy of out_proj's products and dq / dk / dv only as the G-form
 * copy of in_proj's (clip/model.py:185-192: nn.MultiheadAttention, its output projection and packed in-projection); the pair-form kernels
 * write those copies - [rows, 3 seg3] bf16, bit for bit leccr_split3 of the f32 tensor - and the f32 tensors are never written.
 * leccr_attn_s3_takes: 1 when the problem runs on those kernels (f32, head size 64, no causal mask, row strides % 4 == 0, copy strides
 * % 8 == 0, LECCR_ATTN_F32 != exact); otherwise use leccr_attn_fwd / _bwd and leccr_split3.
 * Every copy pointer is at its tensor's column 0 inside segment 0; o3 (forward) is form 0 (hi | hi | lo), dq3 / dk3 / dv3 form 1 (hi | lo | hi). */
int leccr_attn_s3_takes(int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv,
                        int64_t ld3, int64_t seg3);
int leccr_attn_fwd_s3(const void* q, const void* k, const void* v, void* o3, float* lse, const uint8_t* kmask,
                      int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv,
                      int64_t ld3, int64_t seg3, float drop_p, uint64_t seed, void* stream);
int leccr_attn_bwd_s3(const void* q, const void* k, const void* v, const void* o3, int64_t ldo3, int64_t sego3, const void* dout,
                      const float* lse, const uint8_t* kmask, void* dq3, void* dk3, void* dv3, float* delta,
                      int64_t b, int64_t heads, int64_t Lq, int64_t Lk, int64_t hd, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                      int64_t ld3, int64_t seg3, float* bias_sum /* optional [3 heads hd]: += column sums of [dq | dk | dv] */,
                      float drop_p, uint64_t seed, void* stream);

/* ---- ViT input stage --------------------------------------------------------------------------------------
 * patchify: image [b,3,R,R] f32 NCHW -> patches [b*g*g, Kp] (T), g = floor(R/p) (the R % p edge pixels are dropped, as
 *   by the stride-p convolution: ViT-L/14 at 384), column order (c, py, px) = the
 *   flattening of conv1.weight [d,3,p,p]; columns 3p^2..Kp-1 are zero padding.  (clip/model.py:211,224-226)
 * vit_assemble: tok [b, N, d] = concat(class_embedding, patch_out[b, N-1, d]) + positional_embedding
 *   (clip/model.py:227-228); its backward scatters dtok into dpatch_out (T) and ACCUMULATES dcls [d], dpos [N,d] f32.
 */
int leccr_patchify(const float* image, void* patches, int64_t b, int64_t R, int64_t p, int64_t Kp, int dtype,
                   void* stream);
int leccr_vit_assemble_fwd(const void* patch_out, const float* cls, const float* pos, void* tok, int64_t b,
                           int64_t N, int64_t d, int dtype, void* stream);
int leccr_vit_assemble_bwd(const void* dtok, void* dpatch_out, float* dcls, float* dpos, int64_t b, int64_t N,
                           int64_t d, int dtype, void* stream);

/* ---- BERT embedding stage ---------------------------------------------------------------------------------
 * emb[b,L,:] = word[ids] + position[0..L-1] + token_type[0]    (HF BertEmbeddings, called at xvlm.py:238)
 * bwd ACCUMULATES f32: dword[ids] += demb (rows with id == pad_id skipped: nn.Embedding padding_idx),
 *     dpos[l] += sum_b, dtype0 += sum_{b,l}.
 */
int leccr_bert_embed_fwd(const int64_t* ids, const float* word, const float* pos, const float* type0, void* emb,
                         int64_t b, int64_t L, int64_t d, int64_t vocab, int dtype, void* stream);
int leccr_bert_embed_bwd(const int64_t* ids, const void* demb, float* dword, float* dpos, float* dtype0, int64_t b,
                         int64_t L, int64_t d, int64_t vocab, int64_t pad_id, int dtype, void* stream);

/* ---- small reductions -------------------------------------------------------------------------------------
 * colsum: out[n] += sum_m x[m,n]  (bias gradients).   l2norm: y = x / max(||x||_2, 1e-12) per row
 * (F.normalize(dim=-1), xvlm.py:245-256) and its backward.
 */
int leccr_colsum(const void* x, float* out, int64_t M, int64_t N, int64_t ldx, int dtype, void* stream);
int leccr_l2norm_fwd(const void* x, void* y, float* inv_norm, int64_t rows, int64_t d, int dtype, void* stream);
int leccr_l2norm_bwd(const void* dy, const void* y, const float* inv_norm, void* dx, int64_t rows, int64_t d,
                     int dtype, void* stream);

/* ---- similarity + contrastive losses (f32 in, f32 out) ----------------------------------------------------
 * sim_infonce: get_contrastive_loss (models/xvlm.py:260-292).  a,b [G,D] L2-normalised, idx [G] int64 or NULL.
 *   logits = a b^T / temp ; labels = eq(idx,idx^T)/rowsum (or identity) ; loss = (CE_rows + CE_cols)/2.
 *   `ws` (>= 2*G + 4 + 8*G*ceil(G/64) floats): row / column log-sum-exps (the part the backward reads), then the forward's per-tile
 *   online-softmax partials: ONE f32 matrix-core product serves both directions, reduced 64 x 64 tile by tile along its rows and
 *   its columns - the G x G logits are never in memory (D % 8 != 0 falls back to a row-at-a-time kernel).
 *   loss: 1 float.  Backward gives da, db [G,D] (full gathered rows; the caller keeps its local slice,
 *   xvlm.py:62-67) and dtemp (1 float), all scaled by `gscale`.
 * capsim: get_caption_contrastive_loss (model_retrieval_caption.py:145-152).  c [b,n,D], t [b,D]:
 *   logits[i,j] = max_n <c[i,n], t[j]> / temp, hard labels.  argmax [b,b] int32 and ws (>= b*b + 2*b floats:
 *   logits + row/column log-sum-exps) are saved for backward.
 * dstl: dstl_loss (model_retrieval_caption.py:94-116): KL(softmax(a*norm(s v^T) + (1-a)*norm(max_n c s^T)) ||
 *   softmax(t v^T)) batchmean; gradients flow to t and v only (labels detached).  ws >= 3*G*G + G + 8 floats.
 * cvpool: caption_vision_loss (model_retrieval_caption.py:118-143) in pooled form: the mean over both token axes
 *   is linear, so sim[i,j] = <mean_c chat_i, mean_v vhat_j>; chat/vhat are normalised over the TOKEN axis
 *   (F.normalize default dim=1).  cvpool_fwd: x [b, L, d] (projected tokens, T) -> pooled [b, d] f32 +
 *   inv_norm [b, d] f32; cvpool_bwd: dpooled -> dx.   The G x G soft-label CE on the pooled vectors is
 *   sim_infonce_rows (one direction, no temperature).
 * capreg: caption_regularization (model_retrieval_caption.py:65-73): mean(c_hat c_hat^T - I), c [b,n,d].
 */
/* temp_dev (the four temperature-taking calls): NULL = use the host value `temp`; else a DEVICE pointer to the temperature
 * (the learnable scalar XVLMBase.temp, models/xvlm.py:177), read by the kernels at run time - `temp` is then ignored and the
 * training step needs no host read of a device scalar (the reference's `/ self.temp` is a device op too). */
int leccr_sim_infonce_fwd(const float* a, const float* b, const int64_t* idx, float temp, int64_t G, int64_t D,
                          int both_dirs, float* loss, float* ws, const float* temp_dev, void* stream);
int leccr_sim_infonce_bwd(const float* a, const float* b, const int64_t* idx, float temp, int64_t G, int64_t D,
                          int both_dirs, const float* ws, float gscale, float* da, float* db, float* dtemp,
                          int64_t row0, int64_t nrows, const float* temp_dev, void* stream);
int leccr_capsim_fwd(const float* c, const float* t, float temp, int64_t b, int64_t n, int64_t D, float* loss,
                     int32_t* argmax, float* ws, const float* temp_dev, void* stream);
int leccr_capsim_bwd(const float* c, const float* t, float temp, int64_t b, int64_t n, int64_t D,
                     const int32_t* argmax, const float* ws, float gscale, float* dc, float* dt, float* dtemp,
                     const float* temp_dev, void* stream);
int leccr_dstl_fwd(const float* v, const float* c, const float* s, const float* t, float alpha, int64_t G,
                   int64_t n, int64_t D, float* loss, float* ws, void* stream);
int leccr_dstl_bwd(const float* v, const float* t, int64_t G, int64_t D, const float* ws, float gscale, float* dv,
                   float* dt, int64_t row0, int64_t nrows, void* stream);
/* sim_infonce_bwd / dstl_bwd write the gradient rows [row0, row0 + nrows) only (and sim_infonce_bwd adds only those rows'
 * share to dtemp): under data parallelism the gathered operands' gradient is kept for the rank's own rows alone
 * (AllGather backward, models/xvlm.py:62-67), so a rank passes its shard and skips (W-1)/W of the backward work.
 * (0, G) = everything. */
int leccr_cvpool_fwd(const void* x, float* pooled, float* inv_norm, int64_t b, int64_t L, int64_t d, int dtype,
                     void* stream);
int leccr_cvpool_bwd(const void* x, const float* inv_norm, const float* dpooled, void* dx, int64_t b, int64_t L,
                     int64_t d, int dtype, void* stream);
int leccr_capreg_fwd(const void* c, float* loss, int64_t b, int64_t n, int64_t d, int dtype, void* stream);
int leccr_capreg_bwd(const void* c, float gscale, void* dc, int64_t b, int64_t n, int64_t d, int dtype,
                     void* stream);

/* ---- evaluation -------------------------------------------------------------------------------------------
 * rank_of_gt: for each row r of scores [R,C] f32 and each of its ground-truth columns gt[r*per .. r*per+per-1]
 * (int32, -1 = unused), rank = number of columns with a strictly larger score; out[r] = min over its gts.
 * Equals np.where(np.argsort(row)[::-1] == gt) for tie-free rows (image_Retrieval_caption.py:265-292).
 */
int leccr_rank_of_gt(const float* scores, const int32_t* gt, int32_t* out, int64_t R, int64_t C, int64_t ldc,
                     int64_t per, void* stream);

/* ---- optimizer / parameter arena --------------------------------------------------------------------------
 * adamw: decoupled weight decay AdamW exactly as transformers.AdamW (optim.py:1,58-63): p -= lr*wd*p after the
 *   Adam update is NOT what HF does - HF applies  p += -step_size * m/denom  then  p += -lr*wd*p ; bias
 *   correction on.  One launch over a flat f32 arena; `wd_mask` [n_seg] gives per-segment weight decay via
 *   seg_end offsets (sorted, cumulative, in elements).
 *   lp_bf16 (optional, n bf16): the bf16 compute copy of the updated parameters, written in the same pass;
 *   seg_wd < 0 marks a segment the optimizer skips (its bf16 copy is still refreshed).
 * cast_bf16: dst_bf16[i] = src_f32[i];  transpose_bf16: dst[c,r] = bf16(src[r,c]) for an [R,C] f32 matrix.
 */
int leccr_adamw(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                const float* seg_wd, int64_t n_seg, float lr, float beta1, float beta2, float eps, int64_t step,
                float gscale, void* lp_bf16, void* stream);
int leccr_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
int leccr_transpose_cast(const float* src, void* dst, int64_t R, int64_t C, int dtype, void* stream);
/* Many transposes in one launch.  table: n rows of 5 int64 {first tile index (cumulative, 64x64 tiles, row-major over
 * ceil(R/64) x ceil(C/64)), src pointer (f32 [R,C]), dst pointer (bf16 [C,R]), R, C}, in device memory. */
int leccr_transpose_cast_batched(const int64_t* table, int64_t n, int64_t total_tiles, void* stream);
int leccr_axpy(const void* x, void* y, float a, int64_t n, int dtype, void* stream);   /* y += a*x */
/* Elementwise dropout y[i] = keep(seed, i) ? x[i] / (1-p) : 0 (nn.Dropout at attention.py:29-37,92-100; HF
 * hidden_dropout_prob).  The same call on the incoming gradient is the backward.  y may alias x. */
int leccr_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, int dtype, void* stream);

/* ---- fp8 GEMM operands (BASELINE configs[4]: "fp8 MFMA") --------------------------------------------------------
 * Per-tensor scaled OCP fp8 copies of the bf16 GEMM operands, delayed scaling.  A quantisation SITE is 4 floats of device
 * memory {amax, scale, 1/scale, format (0 = e4m3, max 448; 1 = e5m2, max 57344)}:
 *   fp8_amax     : site.amax = max(site.amax, max |x|)                      (first use of a site: scale it from the data itself)
 *   fp8_update   : for each of n sites with amax > 0: scale = 2^floor(log2(fmax / amax)), 1/scale; amax = 0   (once per step)
 *   fp8_quantize : y = fp8(clamp(x * site.scale)) row by row (ldx, ldy in elements; fmt = the site's format), and site.amax accumulates max |x| for the
 *                  NEXT step's scale.  x is bf16 (dtype LECCR_BF16) or f32.
 * leccr_gemm then takes the 1-byte operands with ab_dtype = LECCR_FP8 / LECCR_FP8_BF8A and alpha_dev = {&siteA[2], &siteB[2]}.
 * LayerNorm, softmax, the residual stream, every reduction and the losses stay bf16 / f32 as in the bf16 path.
 */
int leccr_fp8_amax(const void* x, int64_t rows, int64_t cols, int64_t ldx, int dtype, float* site, void* stream);
int leccr_fp8_update(float* sites, int64_t n, void* stream);
/* The same with the running maximum of the NEXT step starting at keep x this step's amax instead of 0 (0 <= keep < 1): the per-step update of
 * the activation / gradient sites.  Producers issue their same-address amax atomic only when their own maximum exceeds the stored one; from 0
 * every resident wave of a site's first launch does (86 us in front of a 171 us LayerNorm pass).  Recorded amax = max(keep x previous, own):
 * never below the true maximum. */
int leccr_fp8_update_keep(float* sites, int64_t n, float keep, void* stream);
int leccr_fp8_quantize(const void* x, void* y, int64_t rows, int64_t cols, int64_t ldx, int64_t ldy, int dtype, int fmt,
                       float* site, void* stream);
/* Weights (current scaling, once per optimizer step): every Linear's bf16 compute copy [R, C] -> its e4m3 copies in both layouts
 * ([R, C] for the forward product, [C, R] for dX) under a scale taken from the weight itself, for ALL weights in three launches.
 * table: n rows of 7 int64 in device memory {first tile (cumulative count of 64 x 64 tiles), bf16 source, e4m3 [R, C], e4m3 [C, R], R, C,
 * site pointer}; R, C multiples of 64, contiguous tensors, site.amax == 0 on entry (as fp8_update leaves it) and on return. */
int leccr_fp8_requantize_weights(const int64_t* table, int64_t n, int64_t total_tiles, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LECCR_HIP_H */
