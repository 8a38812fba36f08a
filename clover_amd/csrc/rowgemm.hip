// Row-streaming MFMA GEMM for the token-parallel projections of the Swin stages (QKV, proj, fc1, fc2
// and their input-gradient GEMMs):
//        Y[M][N] = epilogue( prologue(X)[M][K] * Wt[N][K]^T + bias )
// with M = 10^4..10^5 tokens and K, N of a few hundred, i.e. HBM-bound (AI = K N / (K + N) flop/B is
// below the 312 flop/B ridge).  A library GEMM reaches ~2 TB/s on these shapes and needs separate
// LayerNorm / GELU passes; here
//   * the weight tile (<= 128 x K) sits in LDS for the whole launch, X rows stream through registers
//     exactly once per column tile as MFMA A-fragments (16-B loads);
//   * prologue (optional): x = a + r (residual add, written back as the new residual stream) and
//     row standardisation (x - mean) * rstd — LayerNorm with its affine part folded into Wt / bias by
//     the caller — from the very fragments the MFMA consumes (statistics also written out);
//   * epilogue: + bias, optional erf-GELU (also emitting the pre-activation for the backward) or
//     multiplication by gelu'(pre) (the GELU backward fused into the fc2 input-gradient GEMM);
//   * results leave through a per-wave LDS tile as full 16-B row chunks.
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int RG_THREADS = 256;
constexpr int RG_WAVES = 4;

enum { EPI_NONE = 0, EPI_GELU = 1, EPI_GELU_BWD = 2 };

template <int KS>
struct RGCfg {
    static constexpr int K = KS * 32;
    // columns per workgroup.  K <= 128 (stage-0 widths: LN + QKV, fc1 + GELU, fc2-dgrad + GELU'): 64, so that weight
    // tile + per-wave staging stay near 30 KB and 4-5 workgroups share a CU (+1.0 % on the step against 128, where
    // 61 KB allowed two); K = 192..384: 128 (64 measured equal); wider K: 64 (LDS)
    static constexpr int TN = (KS <= 12 && KS > 4) ? 128 : 64;
    static constexpr int LDW = K + 8;                       // padded LDS row of the weight tile
    static constexpr int LDO = TN + 8;
};

template <int KS, bool STD, int EPI, int NW = RG_WAVES>
__global__ void __launch_bounds__(64 * NW, (NW == 8 && KS <= 4) ? 6 : 1) rowgemm_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ res, bf16_t* __restrict__ sum_out,
    float* __restrict__ mean, float* __restrict__ rstd, bf16_t* __restrict__ xhat_out, const bf16_t* __restrict__ wt,
    const float* __restrict__ bias, const bf16_t* __restrict__ pre_in, bf16_t* __restrict__ y,
    bf16_t* __restrict__ pre_out, int64_t M, int N, int ldx, int ldy, float eps, const float* __restrict__ xscale,
    int rows_per_sample) {
    using C = RGCfg<KS>;
    constexpr int K = C::K, TN = C::TN, LDW = C::LDW, LDO = C::LDO, NT = TN / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ws = reinterpret_cast<bf16_t*>(smem);
    bf16_t* stage = Ws + TN * LDW;                          // [NW][EPI ? 2 : 1][16 * LDO]
    constexpr int NST = (EPI == EPI_NONE) ? 1 : 2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    // 1-D grid of gx * ny workgroups; workgroup i runs on XCD i % 8.  The ny column tiles of one row slice get ids
    // with the same i % 8 that are 8 apart, so they run on the same XCD at about the same time and the rows they
    // all read (x, the residual) come from HBM once and from that L2 afterwards.
    const int ny = (N + TN - 1) / TN;
    const int gx = gridDim.x / ny;
    int bx, by;
    {
        const int i = blockIdx.x, full = (gx / 8) * 8 * ny;           // ids covered by whole groups of 8 row slices
        if (i < full) {
            const int grp = i / (8 * ny), rem = i - grp * 8 * ny;
            by = rem >> 3;
            bx = grp * 8 + (rem & 7);
        } else {                                                       // the gx % 8 leftover row slices: plain order
            const int rem = i - full;
            bx = (gx / 8) * 8 + rem / ny;
            by = rem % ny;
        }
    }
    const int n0 = by * TN;
    const bool first_col = by == 0;

    // ---- weight tile -> LDS (rows >= N zero)
    for (int idx = tid; idx < TN * (K / 8); idx += 64 * NW) {
        const int r = idx / (K / 8), c8 = idx - r * (K / 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n0 + r < N) v = *reinterpret_cast<const uint4*>(wt + (int64_t)(n0 + r) * K + c8 * 8);
        *reinterpret_cast<uint4*>(Ws + r * LDW + c8 * 8) = v;
    }
    // The MFMAs below run with SWAPPED operands (D = W-rows x X-rows), so a lane ends up with 4 CONSECUTIVE output
    // columns (nt*16 + lg*4 + r) of ONE row (lr): the epilogue then packs 4 bf16 into one 8-byte LDS access where the
    // natural orientation needs four 2-byte ones.  bias per lane accordingly: 4 values per column tile.
    // (the TN bias values sit in LDS, one 16-byte read per column tile: held in registers they cost 4 NT VGPRs, which is
    // what kept the LayerNorm-prologue variants from a sixth wave per SIMD)
    float* bias_s = reinterpret_cast<float*>(stage + NW * NST * 16 * LDO);
    for (int n = tid; n < TN; n += 64 * NW) bias_s[n] = (bias && n0 + n < N) ? bias[n0 + n] : 0.f;
    __syncthreads();

    bf16_t* st0 = stage + (wave * NST) * 16 * LDO;
    bf16_t* st1 = st0 + 16 * LDO;
    const int64_t ntiles = (M + 15) / 16;
    for (int64_t tile = (int64_t)bx * NW + wave; tile < ntiles; tile += (int64_t)gx * NW) {
        const int64_t m0 = tile * 16;
        const int64_t row = m0 + lr;
        const bool rv = row < M;
        Frag8 af[KS];
        // ---- A fragments (+ residual, + standardisation)
        if (STD || res) {
            float xs[KS * 8];
            float sum = 0.f;
            // per-sample factor on x (the DropPath factor of the branch whose residual add this prologue performs)
            const float xsc = (xscale && rv) ? xscale[row / rows_per_sample] : 1.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Frag8 a, r;
                a.u4 = rv ? *reinterpret_cast<const uint4*>(x + row * ldx + s * 32 + lg * 8) : make_uint4(0, 0, 0, 0);
                r.u4 = (rv && res) ? *reinterpret_cast<const uint4*>(res + row * ldx + s * 32 + lg * 8)
                                   : make_uint4(0, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = fmaf(bf2f(a.h[e]), xsc, bf2f(r.h[e]));
                    xs[s * 8 + e] = v;
                    sum += v;
                }
                if (res && sum_out && first_col && rv) {
                    Frag8 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o.u[e] = pack2bf(xs[s * 8 + 2 * e], xs[s * 8 + 2 * e + 1]);
                    *reinterpret_cast<uint4*>(sum_out + row * ldx + s * 32 + lg * 8) = o.u4;
                }
            }
            float mu = 0.f, rs = 1.f;
            if (STD) {
                mu = grp4_sum(sum) * (1.0f / K);
                float vs = 0.f;
#pragma unroll
                for (int i = 0; i < KS * 8; ++i) vs += (xs[i] - mu) * (xs[i] - mu);
                rs = rsqrtf(grp4_sum(vs) * (1.0f / K) + eps);
                if (first_col && rv && lg == 0) {
                    mean[row] = mu;
                    rstd[row] = rs;
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    af[s].u[e] = pack2bf((xs[s * 8 + 2 * e] - mu) * rs, (xs[s * 8 + 2 * e + 1] - mu) * rs);
            // the standardised rows, kept for the weight gradient (dW = dY^T x_hat then runs on the LDS-DMA kernel, which
            // cannot transform its operand on the way: 45 us instead of 80 at stage 0)
            if (STD && xhat_out && first_col && rv) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
                    *reinterpret_cast<uint4*>(xhat_out + row * ldx + s * 32 + lg * 8) = af[s].u4;
            }
        } else {
#pragma unroll
            for (int s = 0; s < KS; ++s)
                af[s].u4 = rv ? *reinterpret_cast<const uint4*>(x + row * ldx + s * 32 + lg * 8) : make_uint4(0, 0, 0, 0);
        }
        // ---- GELU backward: stage the pre-activation tile (coalesced) before it is needed per element
        if (EPI == EPI_GELU_BWD) {
            for (int idx = lane; idx < 16 * (TN / 8); idx += 64) {
                const int tr = idx / (TN / 8), c8 = idx - tr * (TN / 8);
                uint4 v = make_uint4(0, 0, 0, 0);
                if (m0 + tr < M && n0 + c8 * 8 < N)
                    v = *reinterpret_cast<const uint4*>(pre_in + (m0 + tr) * ldy + n0 + c8 * 8);
                *reinterpret_cast<uint4*>(st1 + tr * LDO + c8 * 8) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
        // ---- MFMA over the column tiles; epilogue into the wave's staging tile
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Frag8 b;
                b.u4 = *reinterpret_cast<const uint4*>(Ws + (nt * 16 + lr) * LDW + s * 32 + lg * 8);
                acc = mfma16(b, af[s], acc);               // swapped: acc[r] = Y[row lr][col nt*16 + lg*4 + r]
            }
            const int off = lr * LDO + nt * 16 + lg * 4;
            float v[4];
            const float4 bq = *reinterpret_cast<const float4*>(bias_s + nt * 16 + lg * 4);
            v[0] = acc[0] + bq.x; v[1] = acc[1] + bq.y; v[2] = acc[2] + bq.z; v[3] = acc[3] + bq.w;
            if (EPI == EPI_GELU) {
                // pre-activation (kept for the backward)
                *reinterpret_cast<uint2*>(st1 + off) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            } else if (EPI == EPI_GELU_BWD) {
                const uint2 pz = *reinterpret_cast<const uint2*>(st1 + off);
                v[0] *= gelu_erf_grad(bf2f((bf16_t)(pz.x & 0xffff)));
                v[1] *= gelu_erf_grad(bf2f((bf16_t)(pz.x >> 16)));
                v[2] *= gelu_erf_grad(bf2f((bf16_t)(pz.y & 0xffff)));
                v[3] *= gelu_erf_grad(bf2f((bf16_t)(pz.y >> 16)));
            }
            *reinterpret_cast<uint2*>(st0 + off) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int idx = lane; idx < 16 * (TN / 8); idx += 64) {
            const int tr = idx / (TN / 8), c8 = idx - tr * (TN / 8);
            if (m0 + tr < M && n0 + c8 * 8 < N) {
                *reinterpret_cast<uint4*>(y + (m0 + tr) * ldy + n0 + c8 * 8) =
                    *reinterpret_cast<const uint4*>(st0 + tr * LDO + c8 * 8);
                if (EPI == EPI_GELU)
                    *reinterpret_cast<uint4*>(pre_out + (m0 + tr) * ldy + n0 + c8 * 8) =
                        *reinterpret_cast<const uint4*>(st1 + tr * LDO + c8 * 8);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

template <int KS, bool STD, int EPI, int NW = RG_WAVES>
int launch_rg(const void* x, const void* res, void* sum_out, float* mean, float* rstd, void* xhat_out, const void* wt,
              const float* bias, const void* pre_in, void* y, void* pre_out, int64_t M, int N, int ldx, int ldy,
              float eps, hipStream_t st, const float* xscale = nullptr, int rps = 1) {
    using C = RGCfg<KS>;
    constexpr int NST = (EPI == EPI_NONE) ? 1 : 2;
    const size_t lds = (size_t)C::TN * C::LDW * 2 + (size_t)NW * NST * 16 * C::LDO * 2 + (size_t)C::TN * 4;
    if (lds > 160 * 1024) return CLV_ERR_UNSUPPORTED;
    // a weight tile of >= 64 KB leaves ONE workgroup per CU: eight waves share it instead of four (K = 288 / 384: the
    // stage-0 qkv input gradient and fc2)
    if constexpr (NW == RG_WAVES && KS >= 6 && KS <= 12) {
        static const int wv = getenv("CLV_RG_WAVES") ? atoi(getenv("CLV_RG_WAVES")) : 8;
        const size_t wbytes = (size_t)C::TN * C::LDW * 2 + (size_t)C::TN * 4, per_wave = (size_t)NST * 16 * C::LDO * 2;
        if (wbytes >= 48 * 1024) {
            if ((wv == 12 || (KS <= 8 && wv >= 8)) && wbytes + 12 * per_wave <= 160 * 1024)
                return launch_rg<KS, STD, EPI, 12>(x, res, sum_out, mean, rstd, xhat_out, wt, bias, pre_in, y, pre_out, M, N, ldx,
                                                   ldy, eps, st, xscale, rps);
            if (wv >= 8 && wbytes + 8 * per_wave <= 160 * 1024)
                return launch_rg<KS, STD, EPI, 8>(x, res, sum_out, mean, rstd, xhat_out, wt, bias, pre_in, y, pre_out, M, N, ldx,
                                                  ldy, eps, st, xscale, rps);
        }
    }
    // K <= 128 with the GELU' epilogue (fc1's input gradient at stage 0) or the LayerNorm prologue (qkv, fc1): 8-wave
    // workgroups put 24 waves on a CU where five 4-wave ones (LDS) put 20 — 108 -> 91 us for the former.  The prologue
    // variants fit the 80 VGPRs that 24 waves allow since the bias row moved to LDS (K = 96: 76-78; K = 128 spills: 4 waves).
    if constexpr (NW == RG_WAVES && ((KS <= 4 && !STD && EPI == EPI_GELU_BWD) || (KS <= 3 && STD))) {
        static const int w3 = getenv("CLV_RG_WAVES3") ? atoi(getenv("CLV_RG_WAVES3")) : 8;
        if (w3 == 8)
            return launch_rg<KS, STD, EPI, 8>(x, res, sum_out, mean, rstd, xhat_out, wt, bias, pre_in, y, pre_out, M, N, ldx, ldy,
                                              eps, st, xscale, rps);
    }
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_kernel<KS, STD, EPI, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int ny = (N + C::TN - 1) / C::TN;
    const int64_t row_blocks = ((M + 15) / 16 + NW - 1) / NW;
    // persistent workgroups, exactly as many as are resident at once (LDS-limited): every workgroup loads its weight
    // tile (up to 100 KB) ONCE and then streams its share of the rows — with more workgroups than that the tile
    // reload dominates (measured: 1536 workgroups of 2 row tiles per wave ran fc2 at a third of this)
    int occ = (int)((160 * 1024) / lds);
    if (occ < 1) occ = 1;
    if (occ > 4) occ = 4;
    int64_t gx = (256 * occ + ny - 1) / ny;
    if (ny > 3 || EPI != EPI_NONE) {     // many column tiles (small weight tiles), or a VALU-heavy GELU epilogue that
                                         // wants more waves in flight than tile reloads cost: finer row slices
        gx = 1536 / ny;
        if (gx < 256) gx = 256;
    }
    static const int mult = getenv("CLV_RG_MULT") ? atoi(getenv("CLV_RG_MULT")) : 1;   // probe: finer row slices
    gx *= mult;
    if (gx < 8) gx = 8;
    if (gx > row_blocks) gx = row_blocks;
    rowgemm_kernel<KS, STD, EPI, NW><<<dim3((unsigned)(gx * ny)), dim3(64 * NW), lds, st>>>(
        (const bf16_t*)x, (const bf16_t*)res, (bf16_t*)sum_out, mean, rstd, (bf16_t*)xhat_out, (const bf16_t*)wt, bias,
        (const bf16_t*)pre_in, (bf16_t*)y, (bf16_t*)pre_out, M, N, ldx, ldy, eps, xscale, rps > 0 ? rps : 1);
    return clv_check_launch();
}

template <int KS>
int dispatch_rg(bool stdz, int epi, const void* x, const void* res, void* sum_out, float* mean, float* rstd,
                void* xhat_out, const void* wt, const float* bias, const void* pre_in, void* y, void* pre_out, int64_t M, int N,
                int ldx, int ldy, float eps, hipStream_t st, const float* xscale, int rps) {
#define RG_CALL(S, E) launch_rg<KS, S, E>(x, res, sum_out, mean, rstd, xhat_out, wt, bias, pre_in, y, pre_out, M, N, ldx, ldy, eps, st, xscale, rps)
    if (stdz) {
        if constexpr (KS <= 8) {                            // standardisation keeps the fp32 row in registers
            if (epi == EPI_NONE) return RG_CALL(true, EPI_NONE);
            if (epi == EPI_GELU) return RG_CALL(true, EPI_GELU);
        }
        return CLV_ERR_UNSUPPORTED;
    }
    if (epi == EPI_NONE) return RG_CALL(false, EPI_NONE);
    if (epi == EPI_GELU_BWD) return RG_CALL(false, EPI_GELU_BWD);
    return CLV_ERR_UNSUPPORTED;
#undef RG_CALL
}

}  // namespace

extern "C" int clv_rowgemm_supported(int32_t N, int32_t K, int32_t standardise) {
    static const int ks_ok[] = {3, 4, 6, 8, 9, 12, 16, 18, 24};
    if (K % 32 || N % 8 || N <= 0) return 0;
    const int ks = K / 32;
    bool ok = false;
    for (int v : ks_ok) ok |= (v == ks);
    if (!ok) return 0;
    if (standardise && ks > 8) return 0;
    return 1;
}

static int rowgemm_impl(const void* x, const void* res, void* sum_out, float* mean, float* rstd, void* xhat_out,
                        const void* wt, const float* bias, const void* pre_in, void* y, void* pre_out, int64_t M, int32_t N,
                        int32_t K, int32_t ldx, int32_t ldy, int32_t standardise, int32_t epilogue, float eps,
                        const float* xscale, int32_t rows_per_sample, void* stream) {
    if (!x || !wt || !y || M <= 0 || !clv_rowgemm_supported(N, K, standardise) || (ldx & 7) || (ldy & 7) || ldx < K ||
        ldy < N)
        return CLV_ERR_ARG;
    if (standardise && (!mean || !rstd)) return CLV_ERR_ARG;
    if (res && !sum_out) return CLV_ERR_ARG;
    if (epilogue == EPI_GELU && !pre_out) return CLV_ERR_ARG;
    if (epilogue == EPI_GELU_BWD && !pre_in) return CLV_ERR_ARG;
    if (res && !standardise) return CLV_ERR_UNSUPPORTED;
    if (xscale && (!res || rows_per_sample <= 0)) return CLV_ERR_ARG;       // the factor belongs to the residual-add prologue
    hipStream_t st = (hipStream_t)stream;
    const bool sz = standardise != 0;
#define RG_KS(V) case V: return dispatch_rg<V>(sz, epilogue, x, res, sum_out, mean, rstd, xhat_out, wt, bias, pre_in, y, pre_out, M, N, ldx, ldy, eps, st, xscale, rows_per_sample)
    switch (K / 32) {
        RG_KS(3); RG_KS(4); RG_KS(6); RG_KS(8); RG_KS(9); RG_KS(12); RG_KS(16); RG_KS(18); RG_KS(24);
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef RG_KS
}

extern "C" int clv_rowgemm(const void* x, const void* res, void* sum_out, float* mean, float* rstd, void* xhat_out,
                           const void* wt,
                           const float* bias, const void* pre_in, void* y, void* pre_out, int64_t M, int32_t N,
                           int32_t K, int32_t ldx, int32_t ldy, int32_t standardise, int32_t epilogue, float eps,
                           void* stream) {
    return rowgemm_impl(x, res, sum_out, mean, rstd, xhat_out, wt, bias, pre_in, y, pre_out, M, N, K, ldx, ldy, standardise,
                        epilogue, eps, nullptr, 1, stream);
}

extern "C" int clv_rowgemm_xs(const void* x, const void* res, void* sum_out, float* mean, float* rstd, void* xhat_out,
                              const void* wt, const float* bias, const void* pre_in, void* y, void* pre_out, int64_t M,
                              int32_t N, int32_t K, int32_t ldx, int32_t ldy, int32_t standardise, int32_t epilogue,
                              float eps, const float* xscale, int32_t rows_per_sample, void* stream) {
    return rowgemm_impl(x, res, sum_out, mean, rstd, xhat_out, wt, bias, pre_in, y, pre_out, M, N, K, ldx, ldy, standardise,
                        epilogue, eps, xscale, rows_per_sample, stream);
}
