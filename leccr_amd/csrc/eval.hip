// eval.hip - on-device ranking for itm_eval (image_Retrieval_caption.py:261-317): the position of a ground-truth
// column in the descending argsort of its row == the number of strictly larger scores in the row (tie-free rows).
#include "common.h"

namespace leccr {
__global__ __launch_bounds__(256) void rank_of_gt_kernel(const float* __restrict__ scores, const int32_t* __restrict__ gt,
                                                         int32_t* __restrict__ out, int64_t R, int64_t C, int64_t ld, int per) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* row = scores + r * ld;
    int best = 0x7fffffff;
    for (int q = 0; q < per; ++q) {
        const int32_t g = gt[r * per + q];
        if (g < 0 || g >= C) continue;
        const float ref = row[g];
        int cnt = 0;
        for (int64_t j = lane; j < C; j += 64) cnt += row[j] > ref ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        best = cnt < best ? cnt : best;
    }
    if (lane == 0) out[r] = best;
}
}  // namespace leccr

using namespace leccr;

extern "C" int leccr_rank_of_gt(const float* scores, const int32_t* gt, int32_t* out, int64_t R, int64_t C, int64_t ldc,
                                int64_t per, void* stream) {
    LECCR_REQUIRE(scores && gt && out && C > 0 && ldc >= C && per > 0, LECCR_EINVAL, "rank_of_gt: bad argument");
    if (R == 0) return LECCR_OK;
    hipLaunchKernelGGL(rank_of_gt_kernel, dim3((unsigned)cdiv(R, 4)), dim3(256), 0, as_stream(stream), scores, gt, out, R, C, ldc, (int)per);
    return check_launch("rank_of_gt");
}
