// The MLP half of a stage-0 VideoSwin-T block as ONE kernel each way (swin_transformer_3d.py:482-483 norm2, :262-268 Mlp,
// :503 residual): C = 96 channels, 384 hidden units, 10^5 tokens — HBM-bound, and until round 5 five launches forward /
// backward that wrote and re-read the [tokens, 384] hidden tensors nine times (1.4 GB per block and step).
//
//   forward   out = fc2(GELU(fc1(LN(t)))),  t = xscale[b] * a + r          (t is also written: the new residual stream)
//   backward  recomputes xhat = (t - mean) rstd, h = fc1(xhat), GELU(h), GELU'(h) from t; d act = d out W2,
//             d pre = d act GELU'(h), d xhat = d pre W1f, LayerNorm backward (+ the stream's gradient) in the same
//             registers; writes d a / d r and, for the weight-gradient kernels that follow, act, d pre and xhat.
//
// Both weight matrices (2 x 96 x 384 bf16 = 147 KB) sit in LDS for the whole launch (one persistent 8-wave workgroup per
// CU); the hidden activations never leave registers: with the MFMA operands SWAPPED (D = W-rows x token-columns) a lane
// ends up holding, for ONE token, 4 consecutive rows of each 16-row output tile — and when the weight rows of a tile PAIR
// are stored in LDS in the order (g, t, r) -> logical row 32 c + 8 g + 4 t + r  (physical row 32 c + 16 t + 4 g + r),
// those are 8 CONSECUTIVE logical rows (32 c + 8 g .. + 7): exactly the k-slots of the B operand of the next MFMA (the
// hidden units fc2 contracts over) and exactly one 16-byte chunk of a row-major [tokens, 384] / [tokens, 96] tensor.  So
// fc1's accumulators become fc2's operand with a bias + GELU + pack in between, no LDS round trip, and every global
// access is a 16-byte chunk of a token row.  The backward reads W1f^T for d xhat from the same [384][96] LDS image with the
// transposing read (ds_read_b64_tr_b16), so it needs two images as well (W1f, W2^T).
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int FM_C = 96, FM_H = 384, FM_KS = FM_C / 32, FM_CH = FM_H / 32, FM_MT = FM_C / 16;
constexpr int FM_LD1 = FM_C + 8;        // LDS row of an image with 96 columns (208 B: 16 consecutive rows hit 64 distinct banks)
constexpr int FM_LD2 = FM_H + 8;        // LDS row of the [96][384] image (784 B: same)
constexpr int FM_NW = 8;
#ifndef FM_BWD_UNROLL
#define FM_BWD_UNROLL 2
#endif

// physical LDS row p (0..383) of a [384][*] image -> logical hidden unit (see the header): the tiles of a pair interleave
// in groups of 4 so that a lane's D rows of the pair are 8 consecutive units
__device__ __forceinline__ int fm_hidden_of(int p) {
    const int c = p >> 5, t = (p >> 4) & 1, m = p & 15;
    return c * 32 + (m >> 2) * 8 + t * 4 + (m & 3);
}
// physical LDS row p (0..95) of the [96][384] image (forward fc2: rows = output channels) -> logical channel
__device__ __forceinline__ int fm_chan_of(int p) {
    const int mt = p >> 4, m = p & 15;
    return (mt >> 1) * 32 + (m >> 2) * 8 + (mt & 1) * 4 + (m & 3);
}

typedef short fm_v4s_t __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16 (see attention.hip tr4): for this lane, img[row0 + 0..3][c0 + (lane & 15)]
__device__ __forceinline__ uint2 fm_tr4(const bf16_t* base, int LD, int row0, int c0, int lr) {
    const bf16_t* p = base + (row0 + (lr >> 2)) * LD + c0 + (lr & 3) * 4;
    const fm_v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fm_v4s_t*)p);
    union { fm_v4s_t v; uint2 u; } cv;
    cv.v = r;
    return cv.u;
}

struct FmRange {
    int64_t row0, row1;     // this workgroup's token rows
};
__device__ __forceinline__ FmRange fm_range(int64_t M) {
    // contiguous, 16-aligned row ranges, one per workgroup
    const int64_t tiles = (M + 15) / 16, nb = gridDim.x;
    const int64_t t0 = tiles * blockIdx.x / nb, t1 = tiles * (blockIdx.x + 1) / nb;
    FmRange r;
    r.row0 = t0 * 16;
    r.row1 = t1 * 16 < M ? t1 * 16 : M;
    return r;
}

// The raw rows of one 16-token tile (lane: token lr, channels 32 s + 8 lg .. + 7 of the branch and of the residual stream):
// loaded one group AHEAD of their LayerNorm prologue, so that the HBM round trip lies under the previous group's GEMMs.
struct FmRaw {
    uint4 x[FM_KS], r[FM_KS];
    float xsc;
};
__device__ __forceinline__ void fm_load_raw(const bf16_t* __restrict__ a, const bf16_t* __restrict__ res,
                                            const float* __restrict__ xscale, int rows_per_sample, int64_t row, bool rv,
                                            int lg, FmRaw& w) {
#pragma unroll
    for (int s = 0; s < FM_KS; ++s) {
        w.x[s] = rv ? *reinterpret_cast<const uint4*>(a + row * FM_C + s * 32 + lg * 8) : make_uint4(0, 0, 0, 0);
        w.r[s] = (rv && res) ? *reinterpret_cast<const uint4*>(res + row * FM_C + s * 32 + lg * 8) : make_uint4(0, 0, 0, 0);
    }
    w.xsc = (xscale && rv) ? xscale[row / rows_per_sample] : 1.f;
}

// LayerNorm prologue of one 16-token tile: t = xsc * a + r (written to sum_out), statistics, standardised rows as MFMA B
// fragments (lane: token lr, channels 32 s + 8 lg .. + 7).
__device__ __forceinline__ void fm_prologue(const FmRaw& w, bool has_res, bf16_t* __restrict__ sum_out,
                                            float* __restrict__ mean, float* __restrict__ rstd, int64_t row, bool rv, int lg,
                                            float eps, Frag8 (&af)[FM_KS]) {
    float xs[FM_KS * 8];
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < FM_KS; ++s) {
        Frag8 x, r;
        x.u4 = w.x[s];
        r.u4 = w.r[s];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = fmaf(bf2f(x.h[e]), w.xsc, bf2f(r.h[e]));
            xs[s * 8 + e] = v;
            sum += v;
        }
        if (has_res && sum_out && rv) {
            Frag8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.u[e] = pack2bf(xs[s * 8 + 2 * e], xs[s * 8 + 2 * e + 1]);
            *reinterpret_cast<uint4*>(sum_out + row * FM_C + s * 32 + lg * 8) = o.u4;
        }
    }
    const float mu = grp4_sum(sum) * (1.0f / FM_C);
    float vs = 0.f;
#pragma unroll
    for (int i = 0; i < FM_KS * 8; ++i) vs += (xs[i] - mu) * (xs[i] - mu);
    const float rs = rsqrtf(grp4_sum(vs) * (1.0f / FM_C) + eps);
    if (rv && lg == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
#pragma unroll
    for (int s = 0; s < FM_KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) af[s].u[e] = pack2bf((xs[s * 8 + 2 * e] - mu) * rs, (xs[s * 8 + 2 * e + 1] - mu) * rs);
}

// fc1 + GELU + fc2 of one group of NT 16-token tiles (af: their standardised rows) -> out.  The twelve 32-unit chunks of the
// hidden layer are unrolled with the LDS reads of the weight fragments issued AHEAD of their use: W2's fragments of chunk c
// before fc1's MFMAs of chunk c (they are needed after the GELU), W1's of chunk c + 1 before the GELU of chunk c — with two
// waves per SIMD nothing else hides the ~130-cycle LDS round trip between the dependent phases of a chunk.
template <int NT, int ABL>
__device__ __forceinline__ void fm_fwd_group(const bf16_t* __restrict__ W1s, const bf16_t* __restrict__ W2s,
                                             const float* __restrict__ b1s, const float* __restrict__ b2s,
                                             const Frag8 (&af)[2][FM_KS], bf16_t* __restrict__ out, int64_t m0,
                                             int64_t row_end, int lg, int lr) {
    f32x4_t oacc[FM_MT][NT];
#pragma unroll
    for (int mt = 0; mt < FM_MT; ++mt)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) oacc[mt][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16_t* w1p = W1s + lr * FM_LD1 + lg * 8;               // + (32 c + 16 t) rows + 32 s columns
    const bf16_t* w2p = W2s + lr * FM_LD2 + lg * 8;               // + 16 mt rows + 32 c columns
    Frag8 wa[FM_KS][2];
#pragma unroll
    for (int s = 0; s < FM_KS; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) wa[s][t].u4 = *reinterpret_cast<const uint4*>(w1p + (t * 16) * FM_LD1 + s * 32);
#pragma unroll
    for (int c = 0; c < FM_CH; ++c) {
        Frag8 wb[FM_MT];
#pragma unroll
        for (int mt = 0; mt < FM_MT; ++mt) wb[mt].u4 = *reinterpret_cast<const uint4*>(w2p + (mt * 16) * FM_LD2 + c * 32);
        // (the scheduler otherwise sinks every ds_read next to the MFMA that consumes it, behind an s_waitcnt lgkmcnt(0): 24
        // exposed LDS round trips per chunk — 3 200 cycles per chunk where the MFMAs are 400)
        __builtin_amdgcn_sched_barrier(0);
        // ---- fc1: h^T [32 hidden of chunk c][16 NT tokens]; the bias SEEDS the accumulators (a lane's four rows of tile t
        // are hidden units 32 c + 8 lg + 4 t .. + 3)
        f32x4_t h[2][NT];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float4 bq = *reinterpret_cast<const float4*>(b1s + c * 32 + lg * 8 + t * 4);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) h[t][tt] = (f32x4_t){bq.x, bq.y, bq.z, bq.w};
        }
#pragma unroll
        for (int s = 0; s < FM_KS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) h[t][tt] = mfma16(wa[s][t], af[tt][s], h[t][tt]);
        if (c + 1 < FM_CH) {
#pragma unroll
            for (int s = 0; s < FM_KS; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    wa[s][t].u4 = *reinterpret_cast<const uint4*>(w1p + ((c + 1) * 32 + t * 16) * FM_LD1 + s * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GELU; the accumulators of the tile pair ARE the next B operand (hidden 32 c + 8 lg .. + 7)
        Frag8 bfr[NT];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                if (ABL & 1) {              // probe: no GELU
                    bfr[tt].u[t * 2 + 0] = pack2bf(h[t][tt][0], h[t][tt][1]);
                    bfr[tt].u[t * 2 + 1] = pack2bf(h[t][tt][2], h[t][tt][3]);
                } else {
                    const f32x2_t g01 = gelu_fast2((f32x2_t){h[t][tt][0], h[t][tt][1]});
                    const f32x2_t g23 = gelu_fast2((f32x2_t){h[t][tt][2], h[t][tt][3]});
                    bfr[tt].u[t * 2 + 0] = pack2bf(g01.x, g01.y);
                    bfr[tt].u[t * 2 + 1] = pack2bf(g23.x, g23.y);
                }
            }
        // ---- fc2: out^T [96][tokens] += W2[:, chunk c] act^T
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < FM_MT; ++mt)
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) oacc[mt][tt] = mfma16(wb[mt], bfr[tt], oacc[mt][tt]);
    }
    // ---- + b2, 16-byte chunks of the token rows (tile pair 2 s, 2 s + 1 -> channels 32 s + 8 lg .. + 7)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) {
        const int64_t row = m0 + tt * 16 + lr;
        if (row < row_end) {
#pragma unroll
            for (int s = 0; s < FM_KS; ++s) {
                const float4 q0 = *reinterpret_cast<const float4*>(b2s + s * 32 + lg * 8);
                const float4 q1 = *reinterpret_cast<const float4*>(b2s + s * 32 + lg * 8 + 4);
                const f32x4_t x0 = oacc[2 * s][tt], x1 = oacc[2 * s + 1][tt];
                uint4 o;
                o.x = pack2bf(x0[0] + q0.x, x0[1] + q0.y);
                o.y = pack2bf(x0[2] + q0.z, x0[3] + q0.w);
                o.z = pack2bf(x1[0] + q1.x, x1[1] + q1.y);
                o.w = pack2bf(x1[2] + q1.z, x1[3] + q1.w);
                *reinterpret_cast<uint4*>(out + row * FM_C + s * 32 + lg * 8) = o;
            }
        }
    }
}

template <int ABL, int NW>
__global__ void __launch_bounds__(64 * NW, 1) mlp96_fwd_kernel(
    const bf16_t* __restrict__ a, const bf16_t* __restrict__ res, bf16_t* __restrict__ sum_out, float* __restrict__ mean,
    float* __restrict__ rstd, const bf16_t* __restrict__ w1f, const float* __restrict__ b1f,
    const bf16_t* __restrict__ w2, const float* __restrict__ b2, bf16_t* __restrict__ out, int64_t M, float eps,
    const float* __restrict__ xscale, int rows_per_sample) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* W1s = reinterpret_cast<bf16_t*>(smem);                   // [384 physical rows][96 (+8)]
    bf16_t* W2s = W1s + FM_H * FM_LD1;                               // [96 physical rows][384 (+8)]
    float* b1s = reinterpret_cast<float*>(W2s + FM_C * FM_LD2);      // [384] logical order
    float* b2s = b1s + FM_H;                                         // [96] logical order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;

    for (int idx = tid; idx < FM_H * (FM_C / 8); idx += 64 * NW) {
        const int p = idx / (FM_C / 8), c8 = idx - p * (FM_C / 8);
        *reinterpret_cast<uint4*>(W1s + p * FM_LD1 + c8 * 8) =
            *reinterpret_cast<const uint4*>(w1f + (int64_t)fm_hidden_of(p) * FM_C + c8 * 8);
    }
    for (int idx = tid; idx < FM_C * (FM_H / 8); idx += 64 * NW) {
        const int p = idx / (FM_H / 8), c8 = idx - p * (FM_H / 8);
        *reinterpret_cast<uint4*>(W2s + p * FM_LD2 + c8 * 8) =
            *reinterpret_cast<const uint4*>(w2 + (int64_t)fm_chan_of(p) * FM_H + c8 * 8);
    }
    for (int n = tid; n < FM_H; n += 64 * NW) b1s[n] = b1f[n];
    for (int n = tid; n < FM_C; n += 64 * NW) b2s[n] = b2 ? b2[n] : 0.f;
    __syncthreads();

    const FmRange rg = fm_range(M);
    const int64_t ntl = (rg.row1 - rg.row0 + 15) / 16;              // 16-token tiles of this workgroup
    const int64_t ngrp = (ntl + 1) / 2;                              // groups of two tiles (the last may be half)
    FmRaw raw[2];
    if (wave < ngrp) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int64_t row = rg.row0 + (int64_t)wave * 32 + tt * 16 + lr;
            fm_load_raw(a, res, xscale, rows_per_sample, row, row < rg.row1, lg, raw[tt]);
        }
    }
    for (int64_t g = wave; g < ngrp; g += NW) {
        const int64_t m0 = rg.row0 + g * 32;
        Frag8 af[2][FM_KS];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int64_t row = m0 + tt * 16 + lr;
            fm_prologue(raw[tt], res != nullptr, sum_out, mean, rstd, row, row < rg.row1, lg, eps, af[tt]);
        }
        if (g + NW < ngrp) {                   // the next group's rows: in flight under this group's GEMMs
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int64_t row = m0 + (int64_t)NW * 32 + tt * 16 + lr;
                fm_load_raw(a, res, xscale, rows_per_sample, row, row < rg.row1, lg, raw[tt]);
            }
        }
        // a half group (the workgroup's odd last tile) runs the one-tile body: half the MFMAs and GELUs
        if (m0 + 16 < rg.row1)
            fm_fwd_group<2, ABL>(W1s, W2s, b1s, b2s, af, out, m0, rg.row1, lg, lr);
        else
            fm_fwd_group<1, ABL>(W1s, W2s, b1s, b2s, af, out, m0, rg.row1, lg, lr);
    }
}

// D-layout registers of a tile pair (X: tile 2 s, Y: tile 2 s + 1; lane group g holds columns 16 mt + 4 g .. + 3 of token
// lr) -> the row-major layout (lane group g: columns 32 s + 8 g .. + 7 as X' (first four), Y' (last four)):
// rows of 16 lanes  X = [X0 X1 X2 X3], Y = [Y0 Y1 Y2 Y3]  ->  X' = [X0 X2 Y0 Y2], Y' = [X1 X3 Y1 Y3].
// v_permlane32_swap: X = [X0 X1 Y0 Y1], Y = [X2 X3 Y2 Y3]; then v_permlane16_swap (odd rows of the first with even rows
// of the second): X = [X0 X2 Y0 Y2], Y = [X1 X3 Y1 Y3].
typedef unsigned fm_u2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void fm_d_to_rows(float& x, float& y) {
    fm_u2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    r = __builtin_amdgcn_permlane16_swap(r.x, r.y, false, false);
    x = __uint_as_float(r.x);
    y = __uint_as_float(r.y);
}

// Backward.  LDS: W1f [384 physical rows][96] (fc1 recompute: plain reads; d xhat: transposing reads) and W2^T [384
// physical rows][96] (d act).
template <int ABL>
__global__ void __launch_bounds__(64 * FM_NW, 1) mlp96_bwd_kernel(
    const bf16_t* __restrict__ tsum, const float* __restrict__ mean, const float* __restrict__ rstd,
    const bf16_t* __restrict__ dout, const bf16_t* __restrict__ dsum, const bf16_t* __restrict__ w1f,
    const float* __restrict__ b1f, const bf16_t* __restrict__ w2t, bf16_t* __restrict__ da, bf16_t* __restrict__ dres,
    bf16_t* __restrict__ act_out, bf16_t* __restrict__ dpre_out, bf16_t* __restrict__ xhat_out, int64_t M,
    const float* __restrict__ xscale, int rows_per_sample) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* W1s = reinterpret_cast<bf16_t*>(smem);                   // [384 physical rows][96 (+8)]
    bf16_t* W2t = W1s + FM_H * FM_LD1;                               // [384 physical rows][96 (+8)]
    float* b1s = reinterpret_cast<float*>(W2t + FM_H * FM_LD1);      // [384] logical order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;

    for (int idx = tid; idx < FM_H * (FM_C / 8); idx += 64 * FM_NW) {
        const int p = idx / (FM_C / 8), c8 = idx - p * (FM_C / 8);
        const int64_t hrow = fm_hidden_of(p);
        *reinterpret_cast<uint4*>(W1s + p * FM_LD1 + c8 * 8) = *reinterpret_cast<const uint4*>(w1f + hrow * FM_C + c8 * 8);
        *reinterpret_cast<uint4*>(W2t + p * FM_LD1 + c8 * 8) = *reinterpret_cast<const uint4*>(w2t + hrow * FM_C + c8 * 8);
    }
    for (int n = tid; n < FM_H; n += 64 * FM_NW) b1s[n] = b1f[n];
    __syncthreads();

    const FmRange rg = fm_range(M);
    const int64_t ntl = (rg.row1 - rg.row0 + 15) / 16;
    const int64_t ngrp = (ntl + 1) / 2;
    for (int64_t g = wave; g < ngrp; g += FM_NW) {
        const int64_t m0 = rg.row0 + g * 32;
        Frag8 xf[2][FM_KS], df[2][FM_KS];        // xhat and d out as B operands (token lr, channels 32 s + 8 lg .. + 7)
        float mu[2], rs[2];
        {
            // ALL loads of the group first, ONE wait: vmcnt retires in order, and the previous group's ~100 act / d pre
            // stores are still in the queue — every separate load-then-wait pair drained that queue again (the compiler
            // interleaves them with their consumers to save registers: 12 drains per group, 55 % of the wave cycles in
            // s_waitcnt)
            uint4 rt[2][FM_KS];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int64_t row = m0 + tt * 16 + lr;
                const bool rv = row < rg.row1;
                mu[tt] = rv ? mean[row] : 0.f;
                rs[tt] = rv ? rstd[row] : 0.f;
#pragma unroll
                for (int s = 0; s < FM_KS; ++s) {
                    rt[tt][s] = rv ? *reinterpret_cast<const uint4*>(tsum + row * FM_C + s * 32 + lg * 8) : make_uint4(0, 0, 0, 0);
                    df[tt][s].u4 = rv ? *reinterpret_cast<const uint4*>(dout + row * FM_C + s * 32 + lg * 8)
                                      : make_uint4(0, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int64_t row = m0 + tt * 16 + lr;
                const bool rv = row < rg.row1;
#pragma unroll
                for (int s = 0; s < FM_KS; ++s) {
                    Frag8 t;
                    t.u4 = rt[tt][s];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        xf[tt][s].u[e] = pack2bf((bf2f(t.h[2 * e]) - mu[tt]) * rs[tt], (bf2f(t.h[2 * e + 1]) - mu[tt]) * rs[tt]);
                    if (rv && xhat_out) *reinterpret_cast<uint4*>(xhat_out + row * FM_C + s * 32 + lg * 8) = xf[tt][s].u4;
                }
            }
        }
        f32x4_t gacc[FM_MT][2];                   // d xhat^T [96][32 tokens], standard tile order
#pragma unroll
        for (int mt = 0; mt < FM_MT; ++mt)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) gacc[mt][tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // W1f and W2^T fragments of chunk 0; inside the loop the reads of chunk c + 1 (and the transposing reads of chunk c)
        // are issued right after the MFMAs that consumed the previous ones, so that they land under the GELU arithmetic
        const bf16_t* w1p = W1s + lr * FM_LD1 + lg * 8;
        const bf16_t* w2p = W2t + lr * FM_LD1 + lg * 8;
        Frag8 wa[FM_KS][2], wc[FM_KS][2];
#pragma unroll
        for (int s = 0; s < FM_KS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                wa[s][t].u4 = *reinterpret_cast<const uint4*>(w1p + (t * 16) * FM_LD1 + s * 32);
                wc[s][t].u4 = *reinterpret_cast<const uint4*>(w2p + (t * 16) * FM_LD1 + s * 32);
            }
#pragma unroll FM_BWD_UNROLL
        for (int c = 0; c < FM_CH; ++c) {
            f32x4_t h[2][2], da_[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float4 bq = *reinterpret_cast<const float4*>(b1s + c * 32 + lg * 8 + t * 4);
                h[t][0] = (f32x4_t){bq.x, bq.y, bq.z, bq.w};      // the bias seeds the recomputed pre-activation
                h[t][1] = h[t][0];
                da_[t][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                da_[t][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < FM_KS; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    h[t][0] = mfma16(wa[s][t], xf[0][s], h[t][0]);
                    h[t][1] = mfma16(wa[s][t], xf[1][s], h[t][1]);
                    da_[t][0] = mfma16(wc[s][t], df[0][s], da_[t][0]);
                    da_[t][1] = mfma16(wc[s][t], df[1][s], da_[t][1]);
                }
            __builtin_amdgcn_sched_barrier(0);
            {
                const int cn = c + 1 < FM_CH ? c + 1 : c;          // (the last chunk re-reads its own: no branch in the body)
#pragma unroll
                for (int s = 0; s < FM_KS; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        wa[s][t].u4 = *reinterpret_cast<const uint4*>(w1p + (cn * 32 + t * 16) * FM_LD1 + s * 32);
                        wc[s][t].u4 = *reinterpret_cast<const uint4*>(w2p + (cn * 32 + t * 16) * FM_LD1 + s * 32);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- GELU and GELU' of the recomputed pre-activation; act and d pre as 16-byte chunks of their token rows
            Frag8 pf[2], actf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    f32x2_t a01, g01, a23, g23;
                    gelu_fast_both2((f32x2_t){h[t][tt][0], h[t][tt][1]}, a01, g01);
                    gelu_fast_both2((f32x2_t){h[t][tt][2], h[t][tt][3]}, a23, g23);
                    actf[tt].u[t * 2 + 0] = pack2bf(a01.x, a01.y);
                    actf[tt].u[t * 2 + 1] = pack2bf(a23.x, a23.y);
                    pf[tt].u[t * 2 + 0] = pack2bf(da_[t][tt][0] * g01.x, da_[t][tt][1] * g01.y);
                    pf[tt].u[t * 2 + 1] = pack2bf(da_[t][tt][2] * g23.x, da_[t][tt][3] * g23.y);
                }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int64_t row = m0 + tt * 16 + lr;
                if (row < rg.row1 && !(ABL & 2)) {     // (ABL 2: probe without the act / d pre stores)
                    *reinterpret_cast<uint4*>(act_out + row * FM_H + c * 32 + lg * 8) = actf[tt].u4;
                    *reinterpret_cast<uint4*>(dpre_out + row * FM_H + c * 32 + lg * 8) = pf[tt].u4;
                }
            }
            // ---- d xhat^T += W1f^T[:, chunk c] d pre^T: A fragments by transposing reads of the W1f image (rows = hidden),
            // issued together in front of the MFMAs (their registers are free only now: the GELU temporaries are gone)
            Frag8 wt[FM_MT];
#pragma unroll
            for (int mt = 0; mt < FM_MT; ++mt) {
                wt[mt].u2[0] = fm_tr4(W1s, FM_LD1, c * 32 + lg * 4, mt * 16, lr);
                wt[mt].u2[1] = fm_tr4(W1s, FM_LD1, c * 32 + 16 + lg * 4, mt * 16, lr);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < FM_MT; ++mt) {
                gacc[mt][0] = mfma16(wt[mt], pf[0], gacc[mt][0]);
                gacc[mt][1] = mfma16(wt[mt], pf[1], gacc[mt][1]);
            }
        }
        // ---- LayerNorm backward in the row layout: dt = rstd (g - mean(g) - xhat mean(g xhat)) + d sum
        // (the six d sum loads of the group at once, in front of the arithmetic: one drain of the store queue, not six)
        uint4 dsr[2][FM_KS];
        float xscr[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int64_t row = m0 + tt * 16 + lr;
            const bool rv = row < rg.row1;
            xscr[tt] = (xscale && rv) ? xscale[row / rows_per_sample] : 1.f;
#pragma unroll
            for (int s = 0; s < FM_KS; ++s)
                dsr[tt][s] = (dsum && rv) ? *reinterpret_cast<const uint4*>(dsum + row * FM_C + s * 32 + lg * 8)
                                          : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int64_t row = m0 + tt * 16 + lr;
            const bool rv = row < rg.row1;
            float gx[FM_KS][8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int s = 0; s < FM_KS; ++s) {
                f32x4_t x = gacc[2 * s][tt], y = gacc[2 * s + 1][tt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float xv = x[r], yv = y[r];
                    fm_d_to_rows(xv, yv);
                    gx[s][r] = xv;
                    gx[s][4 + r] = yv;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1 += gx[s][e];
                    s2 = fmaf(gx[s][e], bf2f(xf[tt][s].h[e]), s2);
                }
            }
            s1 = grp4_sum(s1) * (1.0f / FM_C);
            s2 = grp4_sum(s2) * (1.0f / FM_C);
            const float xsc = xscr[tt];
            if (rv) {
#pragma unroll
                for (int s = 0; s < FM_KS; ++s) {
                    Frag8 ds;
                    ds.u4 = dsr[tt][s];
                    float dt[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        dt[e] = fmaf(rs[tt], gx[s][e] - s1 - bf2f(xf[tt][s].h[e]) * s2, bf2f(ds.h[e]));
                    uint4 o;
                    if (dres) {
                        o.x = pack2bf(dt[0], dt[1]); o.y = pack2bf(dt[2], dt[3]);
                        o.z = pack2bf(dt[4], dt[5]); o.w = pack2bf(dt[6], dt[7]);
                        *reinterpret_cast<uint4*>(dres + row * FM_C + s * 32 + lg * 8) = o;
                    }
                    o.x = pack2bf(dt[0] * xsc, dt[1] * xsc); o.y = pack2bf(dt[2] * xsc, dt[3] * xsc);
                    o.z = pack2bf(dt[4] * xsc, dt[5] * xsc); o.w = pack2bf(dt[6] * xsc, dt[7] * xsc);
                    *reinterpret_cast<uint4*>(da + row * FM_C + s * 32 + lg * 8) = o;
                }
            }
        }
    }
}

constexpr size_t FM_LDS_FWD = (size_t)FM_H * FM_LD1 * 2 + (size_t)FM_C * FM_LD2 * 2 + (FM_H + FM_C) * 4;
constexpr size_t FM_LDS_BWD = (size_t)2 * FM_H * FM_LD1 * 2 + FM_H * 4;
static_assert(FM_LDS_FWD <= 160 * 1024 && FM_LDS_BWD <= 160 * 1024, "both weight images must fit the 160 KB of a CU");

int fm_grid(int64_t M) {
    static const int per = getenv("CLV_FMLP_GRID") ? atoi(getenv("CLV_FMLP_GRID")) : 256;     // one workgroup per CU
    const int64_t tiles = (M + 15) / 16;
    return (int)(tiles < per ? tiles : per);
}

}  // namespace

extern "C" int clv_mlp_fused_supported(int32_t C, int32_t hidden) { return C == FM_C && hidden == FM_H; }

extern "C" int clv_mlp_fused_fwd(const void* a, const void* res, void* sum_out, float* mean, float* rstd, const void* w1f,
                                 const float* b1f, const void* w2, const float* b2, void* out, int64_t M, int32_t C,
                                 int32_t hidden, float eps, const float* xscale, int32_t rows_per_sample, void* stream) {
    if (!clv_mlp_fused_supported(C, hidden)) return CLV_ERR_UNSUPPORTED;
    if (!a || !mean || !rstd || !w1f || !b1f || !w2 || !out || M <= 0) return CLV_ERR_ARG;
    if (res && !sum_out) return CLV_ERR_ARG;
    if (xscale && (!res || rows_per_sample <= 0)) return CLV_ERR_ARG;
    if ((((uintptr_t)a) | ((uintptr_t)res) | ((uintptr_t)sum_out) | ((uintptr_t)w1f) | ((uintptr_t)w2) | ((uintptr_t)out)) & 15)
        return CLV_ERR_ARG;
    static const int abl = getenv("CLV_FMLP_ABL") ? atoi(getenv("CLV_FMLP_ABL")) : 0;      // probe builds (wrong results)
    static const int nw = getenv("CLV_FMLP_NW") ? atoi(getenv("CLV_FMLP_NW")) : 8;          // waves per workgroup (= per CU); 12 spills since the row prefetch
    auto kern = abl == 1 ? &mlp96_fwd_kernel<1, 8> : nw == 8 ? &mlp96_fwd_kernel<0, 8> : &mlp96_fwd_kernel<0, 12>;
    const int threads = (abl == 1 || nw == 8) ? 512 : 768;
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)FM_LDS_FWD) == hipSuccess;
    (void)attr;
    hipLaunchKernelGGL(kern, dim3((unsigned)fm_grid(M)), dim3(threads), FM_LDS_FWD, (hipStream_t)stream,
                       (const bf16_t*)a, (const bf16_t*)res, (bf16_t*)sum_out, mean, rstd, (const bf16_t*)w1f, b1f,
                       (const bf16_t*)w2, b2, (bf16_t*)out, M, eps, xscale, rows_per_sample > 0 ? rows_per_sample : 1);
    return clv_check_launch();
}

extern "C" int clv_mlp_fused_bwd(const void* tsum, const float* mean, const float* rstd, const void* dout, const void* dsum,
                                 const void* w1f, const float* b1f, const void* w2t, void* da, void* dres, void* act_out,
                                 void* dpre_out, void* xhat_out, int64_t M, int32_t C, int32_t hidden, const float* xscale,
                                 int32_t rows_per_sample, void* stream) {
    if (!clv_mlp_fused_supported(C, hidden)) return CLV_ERR_UNSUPPORTED;
    if (!tsum || !mean || !rstd || !dout || !w1f || !b1f || !w2t || !da || !act_out || !dpre_out || M <= 0) return CLV_ERR_ARG;
    if (xscale && rows_per_sample <= 0) return CLV_ERR_ARG;
    if ((((uintptr_t)tsum) | ((uintptr_t)dout) | ((uintptr_t)dsum) | ((uintptr_t)w1f) | ((uintptr_t)w2t) | ((uintptr_t)da) |
         ((uintptr_t)dres) | ((uintptr_t)act_out) | ((uintptr_t)dpre_out) | ((uintptr_t)xhat_out)) & 15)
        return CLV_ERR_ARG;
    static const int abl = getenv("CLV_FMLP_ABL") ? atoi(getenv("CLV_FMLP_ABL")) : 0;      // probe builds (wrong results)
    auto kern = abl == 2 ? &mlp96_bwd_kernel<2> : &mlp96_bwd_kernel<0>;
    static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)FM_LDS_BWD) == hipSuccess;
    (void)attr;
    hipLaunchKernelGGL(kern, dim3((unsigned)fm_grid(M)), dim3(64 * FM_NW), FM_LDS_BWD, (hipStream_t)stream,
                       (const bf16_t*)tsum, mean, rstd, (const bf16_t*)dout, (const bf16_t*)dsum, (const bf16_t*)w1f, b1f,
                       (const bf16_t*)w2t, (bf16_t*)da, (bf16_t*)dres, (bf16_t*)act_out, (bf16_t*)dpre_out,
                       (bf16_t*)xhat_out, M, xscale, rows_per_sample > 0 ? rows_per_sample : 1);
    return clv_check_launch();
}
