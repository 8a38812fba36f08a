// Row softmax (+ additive key mask, + attention-probability dropout) and its backward, for the UNFUSED attention
// path that serves sequences whose K/V do not fit LDS in the fused kernels (N > 448 keys: the fusion encoder at
// 32 frames, 16*49 + 32 = 816 tokens).  The two GEMMs either side (Q.K^T, P.V and their gradients) are plain
// batched library GEMMs; these kernels are the part between them (HF BertSelfAttention, transformers 4.6.1
// modeling_bert.py: scores / sqrt(d) + mask -> softmax -> dropout, reached from
// mmaction/models/backbones/cross_transformer.py:95-108 through BertEncoder).
//
// One wave per row, the row held in registers (<= 2048 keys): lane l owns keys l, l+64, ... so every load / store
// instruction of a wave covers 128 contiguous bytes.  HBM-bound: fwd reads 2 B and writes 2 (+2 with dropout) B per
// score, bwd reads 4 and writes 2.
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int SM_WAVES = 4;

template <int ITERS, bool DROP>
__global__ void __launch_bounds__(SM_WAVES * 64)
softmax_rows_fwd_kernel(const bf16_t* __restrict__ scores, const float* __restrict__ kmask, bf16_t* __restrict__ p,
                        bf16_t* __restrict__ pd, const unsigned long long* __restrict__ seed, int64_t rows, int S,
                        int ld, int rows_per_group, float scale_log2e, unsigned thresh, float inv_keep) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * SM_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bf16_t* src = scores + row * ld;
    const float* km = kmask ? kmask + (row / rows_per_group) * S : nullptr;
    float x[ITERS];
    float m = -INFINITY;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int j = it * 64 + lane;
        // exp2 domain: (s * scale + mask) * log2(e)
        x[it] = j < S ? bf2f(src[j]) * scale_log2e + (km ? km[j] * 1.4426950408889634f : 0.f) : -INFINITY;
        m = fmaxf(m, x[it]);
    }
    m = wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        x[it] = __builtin_amdgcn_exp2f(x[it] - m);
        sum += x[it];
    }
    const float inv = 1.0f / wave_sum(sum);
    const unsigned long long sd = DROP ? seed[0] : 0ull;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int j = it * 64 + lane;
        if (j < S) {
            const float v = x[it] * inv;
            p[row * ld + j] = f2bf(v);
            if (DROP) pd[row * ld + j] = f2bf(v * keep_scale(sd, (unsigned)row, (unsigned)j, thresh, inv_keep));
        }
    }
}

// dS = P * (dP - sum_j P_j dP_j) * scale,  dP = dPd * keep/(1-p)
template <int ITERS, bool DROP>
__global__ void __launch_bounds__(SM_WAVES * 64)
softmax_rows_bwd_kernel(const bf16_t* __restrict__ p, const bf16_t* __restrict__ dpd, bf16_t* __restrict__ ds,
                        const unsigned long long* __restrict__ seed, int64_t rows, int S, int ld, float scale,
                        unsigned thresh, float inv_keep) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * SM_WAVES + (threadIdx.x >> 6);
    if (row >= rows) return;
    const unsigned long long sd = DROP ? seed[0] : 0ull;
    float pv[ITERS], dp[ITERS];
    float dot = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int j = it * 64 + lane;
        pv[it] = dp[it] = 0.f;
        if (j < S) {
            pv[it] = bf2f(p[row * ld + j]);
            dp[it] = bf2f(dpd[row * ld + j]);
            if (DROP) dp[it] *= keep_scale(sd, (unsigned)row, (unsigned)j, thresh, inv_keep);
            dot += pv[it] * dp[it];
        }
    }
    dot = wave_sum(dot);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int j = it * 64 + lane;
        if (j < S) ds[row * ld + j] = f2bf(pv[it] * (dp[it] - dot) * scale);
    }
}

int pick_iters(int S) {
    const int need = (S + 63) / 64;
    const int opts[] = {4, 8, 13, 16, 32};
    for (int o : opts) if (o >= need) return o;
    return -1;
}

}  // namespace

#define SM_DISPATCH(KERNEL, DROPV, ...)                                                                            \
    switch (iters) {                                                                                               \
        case 4: hipLaunchKernelGGL((KERNEL<4, DROPV>), grid, dim3(SM_WAVES * 64), 0, st, __VA_ARGS__); break;       \
        case 8: hipLaunchKernelGGL((KERNEL<8, DROPV>), grid, dim3(SM_WAVES * 64), 0, st, __VA_ARGS__); break;       \
        case 13: hipLaunchKernelGGL((KERNEL<13, DROPV>), grid, dim3(SM_WAVES * 64), 0, st, __VA_ARGS__); break;     \
        case 16: hipLaunchKernelGGL((KERNEL<16, DROPV>), grid, dim3(SM_WAVES * 64), 0, st, __VA_ARGS__); break;     \
        default: hipLaunchKernelGGL((KERNEL<32, DROPV>), grid, dim3(SM_WAVES * 64), 0, st, __VA_ARGS__); break;     \
    }

extern "C" int clv_softmax_rows_fwd(const void* scores, const float* kmask, void* p, void* pd, const void* seed,
                                    int64_t rows, int32_t S, int32_t ld, int32_t rows_per_group, float scale,
                                    float dropout_p, void* stream) {
    if (!scores || !p || rows <= 0 || S <= 0 || ld < S || rows_per_group <= 0 || dropout_p < 0.f || dropout_p >= 1.f)
        return CLV_ERR_ARG;
    if (dropout_p > 0.f && (!pd || !seed)) return CLV_ERR_ARG;
    const int iters = pick_iters(S);
    if (iters < 0) return CLV_ERR_UNSUPPORTED;
    if (rows > (int64_t)0xffffffffu) return CLV_ERR_UNSUPPORTED;          // the mask hash takes a 32-bit row id
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + SM_WAVES - 1) / SM_WAVES));
    const float sl = scale * 1.4426950408889634f;
    const unsigned thresh = (unsigned)((double)dropout_p * 4294967296.0);
    const float inv_keep = 1.0f / (1.0f - dropout_p);
    const bf16_t* sc = (const bf16_t*)scores;
    const unsigned long long* sp = (const unsigned long long*)seed;
    if (thresh) {
        SM_DISPATCH(softmax_rows_fwd_kernel, true, sc, kmask, (bf16_t*)p, (bf16_t*)pd, sp, rows, (int)S, (int)ld,
                    (int)rows_per_group, sl, thresh, inv_keep)
    } else {
        SM_DISPATCH(softmax_rows_fwd_kernel, false, sc, kmask, (bf16_t*)p, (bf16_t*)pd, sp, rows, (int)S, (int)ld,
                    (int)rows_per_group, sl, thresh, inv_keep)
    }
    return clv_check_launch();
}

extern "C" int clv_softmax_rows_bwd(const void* p, const void* dpd, void* ds, const void* seed, int64_t rows,
                                    int32_t S, int32_t ld, float scale, float dropout_p, void* stream) {
    if (!p || !dpd || !ds || rows <= 0 || S <= 0 || ld < S || dropout_p < 0.f || dropout_p >= 1.f) return CLV_ERR_ARG;
    if (dropout_p > 0.f && !seed) return CLV_ERR_ARG;
    const int iters = pick_iters(S);
    if (iters < 0) return CLV_ERR_UNSUPPORTED;
    if (rows > (int64_t)0xffffffffu) return CLV_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((rows + SM_WAVES - 1) / SM_WAVES));
    const unsigned thresh = (unsigned)((double)dropout_p * 4294967296.0);
    const float inv_keep = 1.0f / (1.0f - dropout_p);
    const unsigned long long* sp = (const unsigned long long*)seed;
    if (thresh) {
        SM_DISPATCH(softmax_rows_bwd_kernel, true, (const bf16_t*)p, (const bf16_t*)dpd, (bf16_t*)ds, sp, rows, (int)S,
                    (int)ld, scale, thresh, inv_keep)
    } else {
        SM_DISPATCH(softmax_rows_bwd_kernel, false, (const bf16_t*)p, (const bf16_t*)dpd, (bf16_t*)ds, sp, rows, (int)S,
                    (int)ld, scale, thresh, inv_keep)
    }
    return clv_check_launch();
}
