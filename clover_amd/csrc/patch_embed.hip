// PatchEmbed3D (conv3d k=s=(2,4,4) as an MFMA GEMM over non-overlapping patches) fused with
// bias + LayerNorm(C) + the SimMIM-style mask-token blend, writing channels-last bf16 tokens
// for the clean and the masked pass from ONE read of the fp32 clip.
//
// K = 3*2*4*4 = 96 is walked as 3 MFMA k-steps (one per input channel): inside a k-step the
// lane group g = lane>>4 owns (dt, dy-pair) = (g>>1, g&1) and its 8 k-values are two float4
// rows of the clip, so a wave's A-operand load is 16 consecutive tokens x 16 B = 256 B
// contiguous per lane group (coalesced [B,3,T,H,W] reads).  The C x 96 weight lives in
// registers as B-operand fragments for the whole grid-stride loop.
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int PE_THREADS = 256;
constexpr int PE_WAVES = PE_THREADS / 64;

struct PEShape {
    int B, T, H, W;        // input clip dims (T even, H,W multiples of 4)
    int Tp, Hp, Wp;        // token grid
    int mh, mw, ch, cw;    // mask grid and cell size (Hp/mh, Wp/mw)
    int64_t M;             // tokens
};

__device__ __forceinline__ void tok_coords(const PEShape& S, int64_t m, int& b, int& tp, int& hp, int& wp) {
    wp = (int)(m % S.Wp);
    int64_t r = m / S.Wp;
    hp = (int)(r % S.Hp);
    r /= S.Hp;
    tp = (int)(r % S.Tp);
    b = (int)(r / S.Tp);
}

// A-operand fragment (8 k-values of channel c) of token m for lane group g.
__device__ __forceinline__ Frag8 patch_frag(const float* __restrict__ x, const PEShape& S, int64_t m, bool valid,
                                            int c, int g) {
    Frag8 f;
    if (!valid) {
        f.u4 = make_uint4(0, 0, 0, 0);
        return f;
    }
    int b, tp, hp, wp;
    tok_coords(S, m, b, tp, hp, wp);
    const int t = 2 * tp + (g >> 1), y = 4 * hp + 2 * (g & 1);
    const float* p = x + ((((int64_t)b * 3 + c) * S.T + t) * S.H + y) * S.W + 4 * wp;
    const float4 r0 = *reinterpret_cast<const float4*>(p);
    const float4 r1 = *reinterpret_cast<const float4*>(p + S.W);
    f.u[0] = pack2bf(r0.x, r0.y);
    f.u[1] = pack2bf(r0.z, r0.w);
    f.u[2] = pack2bf(r1.x, r1.y);
    f.u[3] = pack2bf(r1.z, r1.w);
    return f;
}

// fp8 variant (BASELINE config 5, "fp8 MFMA ... patch-proj path"): the same 8 k-values per lane, kept in fp32 until the
// token's 96 patch values are all in hand (its row maximum sits in the 4 lane groups x 3 channel steps of lane lr), then
// scaled by 448 / max and converted to OCP e4m3; the product runs on v_mfma_f32_16x16x32_fp8_fp8 (same lane -> k map as
// the bf16 instruction, 8 bytes per lane) and is rescaled by row scale x per-output-channel weight scale afterwards.
__device__ __forceinline__ void patch_frag_f32(const float* __restrict__ x, const PEShape& S, int64_t m, bool valid, int c,
                                               int g, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (!valid) return;
    int b, tp, hp, wp;
    tok_coords(S, m, b, tp, hp, wp);
    const int t = 2 * tp + (g >> 1), y = 4 * hp + 2 * (g & 1);
    const float* p = x + ((((int64_t)b * 3 + c) * S.T + t) * S.H + y) * S.W + 4 * wp;
    const float4 r0 = *reinterpret_cast<const float4*>(p);
    const float4 r1 = *reinterpret_cast<const float4*>(p + S.W);
    v[0] = r0.x; v[1] = r0.y; v[2] = r0.z; v[3] = r0.w; v[4] = r1.x; v[5] = r1.y; v[6] = r1.z; v[7] = r1.w;
}
__device__ __forceinline__ long pack8_fp8(const float (&v)[8], float inv) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4] * inv, v[5] * inv, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6] * inv, v[7] * inv, hi, true);
    return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int C, bool FP8 = false>
__global__ void __launch_bounds__(PE_THREADS) patch_embed_fwd_kernel(
    const float* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mask_token,
    const int64_t* __restrict__ vmask, bf16_t* __restrict__ out_clean, bf16_t* __restrict__ out_masked,
    bf16_t* __restrict__ z_out, float* __restrict__ mean, float* __restrict__ rstd, PEShape S, float eps,
    const unsigned char* __restrict__ w8 = nullptr, const float* __restrict__ wscale = nullptr) {
    constexpr int NT = C / 16, LDO = C + 8;
    __shared__ __attribute__((aligned(16))) bf16_t tile[PE_WAVES][3][16 * LDO];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lg = lane >> 4, lr = lane & 15;
    // weight fragments: B[k = s*32 + lg*8 + j][n = nt*16 + lr] = W[n][k]
    Frag8 wf[3][NT];
    long wf8[3][NT];
    float ws[NT];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (FP8) wf8[s][nt] = *reinterpret_cast<const long*>(w8 + (nt * 16 + lr) * 96 + s * 32 + lg * 8);
            else wf[s][nt].u4 = *reinterpret_cast<const uint4*>(w + (nt * 16 + lr) * 96 + s * 32 + lg * 8);
        }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) ws[nt] = FP8 ? wscale[nt * 16 + lr] : 1.f;
    float bi[NT], ga[NT], be[NT], mt[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        bi[nt] = bias[nt * 16 + lr];
        ga[nt] = gamma ? gamma[nt * 16 + lr] : 1.f;
        be[nt] = beta ? beta[nt * 16 + lr] : 0.f;
        mt[nt] = (out_masked && mask_token) ? mask_token[nt * 16 + lr] : 0.f;
    }
    const int64_t ntiles = (S.M + 15) / 16;
    const int64_t wave = (int64_t)blockIdx.x * PE_WAVES + wv, nwaves = (int64_t)gridDim.x * PE_WAVES;
    const float invC = 1.0f / (float)C;
    for (int64_t tileid = wave; tileid < ntiles; tileid += nwaves) {
        const int64_t m0 = tileid * 16;
        f32x4_t acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if constexpr (FP8) {
            float pv[3][8], amax = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                patch_frag_f32(x, S, m0 + lr, m0 + lr < S.M, s, lg, pv[s]);
#pragma unroll
                for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(pv[s][e]));
            }
            amax = grp4_max(amax);                               // the token's 96 values: 4 lane groups x 3 steps
            const float inv = amax > 0.f ? 448.0f / amax : 1.0f, rsc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const long a8 = pack8_fp8(pv[s], inv);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, wf8[s][nt], acc[nt], 0, 0, 0);
            }
            // acc[nt][r] belongs to token m0 + lg*4 + r: its row scale lives in the lanes with lr = lg*4 + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float sr = __shfl(rsc, lg * 4 + r, 64);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][r] *= sr * ws[nt];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const Frag8 a = patch_frag(x, S, m0 + lr, m0 + lr < S.M, s, lg);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(a, wf[s][nt], acc[nt]);
            }
        }
        // acc[nt][r] = z[token m0 + lg*4 + r][n = nt*16 + lr]  (before bias)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = m0 + lg * 4 + r;
            float zs[NT], sum = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                zs[nt] = acc[nt][r] + bi[nt];
                sum += zs[nt];
            }
            float mu = 0.f, rs = 1.f;
            if (gamma) {
                mu = row16_sum(sum) * invC;
                float vs = 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) vs += (zs[nt] - mu) * (zs[nt] - mu);
                rs = rsqrtf(row16_sum(vs) * invC + eps);
            }
            float wgt = 0.f;
            if (out_masked && m < S.M) {
                int b, tp, hp, wp;
                tok_coords(S, m, b, tp, hp, wp);
                wgt = (float)vmask[((int64_t)b * S.mh + hp / S.ch) * S.mw + wp / S.cw];
            }
            if (m < S.M && lr == 0) {
                if (mean) mean[m] = mu;
                if (rstd) rstd[m] = rs;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float yv = gamma ? (zs[nt] - mu) * rs * ga[nt] + be[nt] : zs[nt];
                const int off = (lg * 4 + r) * LDO + nt * 16 + lr;
                tile[wv][0][off] = f2bf(yv);
                tile[wv][1][off] = f2bf(yv * (1.f - wgt) + mt[nt] * wgt);
                tile[wv][2][off] = f2bf(zs[nt]);
            }
        }
        // the wave's own tile: write full rows, 16 B per lane  (wave-private LDS, no block barrier)
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        constexpr int CH = C / 8;
        for (int idx = lane; idx < 16 * CH; idx += 64) {
            const int tr = idx / CH, c8 = idx - tr * CH;
            const int64_t m = m0 + tr;
            if (m < S.M) {
                if (out_clean)
                    *reinterpret_cast<uint4*>(out_clean + m * C + c8 * 8) =
                        *reinterpret_cast<const uint4*>(&tile[wv][0][tr * LDO + c8 * 8]);
                if (out_masked)
                    *reinterpret_cast<uint4*>(out_masked + m * C + c8 * 8) =
                        *reinterpret_cast<const uint4*>(&tile[wv][1][tr * LDO + c8 * 8]);
                if (z_out)
                    *reinterpret_cast<uint4*>(z_out + m * C + c8 * 8) =
                        *reinterpret_cast<const uint4*>(&tile[wv][2][tr * LDO + c8 * 8]);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// im2col of the clip into bf16 patches [M][96] (k = c*32 + dt*16 + dy*4 + dx), the operand of
// the weight-gradient GEMM dW = dZᵀ · patches.
__global__ void __launch_bounds__(256) im2col_kernel(const float* __restrict__ x, bf16_t* __restrict__ patches,
                                                     PEShape S) {
    // one thread = one (token, c, dt, dy) row of 4 floats -> 4 bf16
    const int64_t total = S.M * 24;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int64_t m = i / 24;
        const int q = (int)(i - m * 24);          // q = c*8 + dt*4 + dy
        const int c = q >> 3, dt = (q >> 2) & 1, dy = q & 3;
        int b, tp, hp, wp;
        tok_coords(S, m, b, tp, hp, wp);
        const float4 v = *reinterpret_cast<const float4*>(
            x + ((((int64_t)b * 3 + c) * S.T + 2 * tp + dt) * S.H + 4 * hp + dy) * S.W + 4 * wp);
        uint2 o;
        o.x = pack2bf(v.x, v.y);
        o.y = pack2bf(v.z, v.w);
        *reinterpret_cast<uint2*>(patches + m * 96 + q * 4) = o;
    }
}

bool make_shape(PEShape& S, int B, int T, int H, int W, int mh, int mw) {
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || (T & 1) || (H & 3) || (W & 3)) return false;
    S.B = B; S.T = T; S.H = H; S.W = W;
    S.Tp = T / 2; S.Hp = H / 4; S.Wp = W / 4;
    S.M = (int64_t)B * S.Tp * S.Hp * S.Wp;
    S.mh = mh > 0 ? mh : 1;
    S.mw = mw > 0 ? mw : 1;
    if (S.Hp % S.mh || S.Wp % S.mw) return false;
    S.ch = S.Hp / S.mh;
    S.cw = S.Wp / S.mw;
    return true;
}

// Backward of the mask-token blend (swin_transformer_3d.py:222-230): out_masked = y (1 - w) + mask_token w and
// out_clean = y, so  dy = d_clean + d_masked (1 - w)  and  d mask_token = sum over tokens of d_masked w
// (w in {0,1} per token, from the [B][mh][mw] video mask).  One pass: each thread owns 8 channels of a row
// stripe; the mask-token sum is folded per block through LDS and added with one atomic per channel.
template <int CV>
__global__ void __launch_bounds__(256) blend_bwd_kernel(const bf16_t* __restrict__ dclean,
                                                        const bf16_t* __restrict__ dmasked,
                                                        const int64_t* __restrict__ vmask, bf16_t* __restrict__ dy,
                                                        float* __restrict__ dmt, PEShape S) {
    constexpr int CH = CV / 8;                    // 16-byte chunks per row
    constexpr int RPB = 256 / CH;                 // rows per block pass
    __shared__ float red[RPB][CV];
    const int t = threadIdx.x, rl = t / CH, ck = t - rl * CH;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (rl < RPB) {
        for (int64_t m = (int64_t)blockIdx.x * RPB + rl; m < S.M; m += (int64_t)gridDim.x * RPB) {
            int b, tp, hp, wp;
            tok_coords(S, m, b, tp, hp, wp);
            const bool masked = dmasked && vmask[((int64_t)b * S.mh + hp / S.ch) * S.mw + wp / S.cw] != 0;
            Frag8 c, d, o;
            c.u4 = dclean ? *reinterpret_cast<const uint4*>(dclean + m * CV + ck * 8) : make_uint4(0, 0, 0, 0);
            d.u4 = dmasked ? *reinterpret_cast<const uint4*>(dmasked + m * CV + ck * 8) : make_uint4(0, 0, 0, 0);
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dm = bf2f(d.h[e]);
                v[e] = bf2f(c.h[e]) + (masked ? 0.f : dm);
                if (masked) acc[e] += dm;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o.u[e] = pack2bf(v[2 * e], v[2 * e + 1]);
            *reinterpret_cast<uint4*>(dy + m * CV + ck * 8) = o.u4;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rl][ck * 8 + e] = acc[e];
    }
    __syncthreads();
    if (dmt && t < CV) {
        float sum = 0.f;
        for (int r = 0; r < RPB; ++r) sum += red[r][t];
        if (sum != 0.f) atomicAdd(dmt + t, sum);
    }
}

}  // namespace

extern "C" int clv_patch_embed_fwd(const float* x, const void* w, const float* bias, const float* gamma,
                                   const float* beta, const float* mask_token, const int64_t* vmask,
                                   void* out_clean, void* out_masked, void* z_out, float* mean, float* rstd,
                                   int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t mh, int32_t mw,
                                   float eps, void* stream) {
    PEShape S;
    if (!x || !w || !bias || (!out_clean && !out_masked)) return CLV_ERR_ARG;
    if ((gamma == nullptr) != (beta == nullptr)) return CLV_ERR_ARG;
    if (out_masked && (!vmask || !mask_token)) return CLV_ERR_ARG;
    if (!make_shape(S, B, T, H, W, out_masked ? mh : 1, out_masked ? mw : 1)) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int64_t ntiles = (S.M + 15) / 16;
    int grid = (int)((ntiles + PE_WAVES - 1) / PE_WAVES);
    if (grid > 1024) grid = 1024;
#define PE_LAUNCH(CV)                                                                                          \
    hipLaunchKernelGGL((patch_embed_fwd_kernel<CV>), dim3(grid), dim3(PE_THREADS), 0, st, x, (const bf16_t*)w, \
                       bias, gamma, beta, mask_token, vmask, (bf16_t*)out_clean, (bf16_t*)out_masked,          \
                       (bf16_t*)z_out, mean, rstd, S, eps)
    switch (C) {
        case 48: PE_LAUNCH(48); break;
        case 96: PE_LAUNCH(96); break;
        case 128: PE_LAUNCH(128); break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef PE_LAUNCH
    return clv_check_launch();
}

extern "C" int clv_patch_embed_fwd_fp8(const float* x, const void* w8, const float* wscale, const float* bias,
                                       const float* gamma, const float* beta, const float* mask_token,
                                       const int64_t* vmask, void* out_clean, void* out_masked, void* z_out, float* mean,
                                       float* rstd, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t mh,
                                       int32_t mw, float eps, void* stream) {
    PEShape S;
    if (!x || !w8 || !wscale || !bias || (!out_clean && !out_masked)) return CLV_ERR_ARG;
    if ((gamma == nullptr) != (beta == nullptr)) return CLV_ERR_ARG;
    if (out_masked && (!vmask || !mask_token)) return CLV_ERR_ARG;
    if (!make_shape(S, B, T, H, W, out_masked ? mh : 1, out_masked ? mw : 1)) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    int64_t ntiles = (S.M + 15) / 16;
    int grid = (int)((ntiles + PE_WAVES - 1) / PE_WAVES);
    if (grid > 1024) grid = 1024;
#define PE_LAUNCH8(CV)                                                                                              \
    hipLaunchKernelGGL((patch_embed_fwd_kernel<CV, true>), dim3(grid), dim3(PE_THREADS), 0, st, x, (const bf16_t*)nullptr, \
                       bias, gamma, beta, mask_token, vmask, (bf16_t*)out_clean, (bf16_t*)out_masked,              \
                       (bf16_t*)z_out, mean, rstd, S, eps, (const unsigned char*)w8, wscale)
    switch (C) {
        case 48: PE_LAUNCH8(48); break;
        case 96: PE_LAUNCH8(96); break;
        case 128: PE_LAUNCH8(128); break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef PE_LAUNCH8
    return clv_check_launch();
}

extern "C" int clv_im2col_patches(const float* x, void* patches, int32_t B, int32_t T, int32_t H, int32_t W,
                                  void* stream) {
    PEShape S;
    if (!x || !patches || !make_shape(S, B, T, H, W, 1, 1)) return CLV_ERR_ARG;
    int64_t total = S.M * 24;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(im2col_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)patches, S);
    return clv_check_launch();
}

extern "C" int clv_patch_embed_blend_bwd(const void* dclean, const void* dmasked, const int64_t* vmask, void* dy,
                                         float* dmask_token, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C,
                                         int32_t mh, int32_t mw, void* stream) {
    PEShape S;
    if (!dy || (!dclean && !dmasked) || (dmasked && (!vmask || !dmask_token))) return CLV_ERR_ARG;
    if (!make_shape(S, B, T, H, W, dmasked ? mh : 1, dmasked ? mw : 1)) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
#define BB_LAUNCH(CV)                                                                                             \
    {                                                                                                             \
        const int rpb = 256 / (CV / 8);                                                                           \
        int64_t g = (S.M + rpb - 1) / rpb;                                                                        \
        if (g > 2048) g = 2048;                                                                                   \
        hipLaunchKernelGGL((blend_bwd_kernel<CV>), dim3((unsigned)g), dim3(256), 0, st, (const bf16_t*)dclean,    \
                           (const bf16_t*)dmasked, vmask, (bf16_t*)dy, dmask_token, S);                           \
    }
    switch (C) {
        case 48: BB_LAUNCH(48) break;
        case 96: BB_LAUNCH(96) break;
        case 128: BB_LAUNCH(128) break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef BB_LAUNCH
    return clv_check_launch();
}
