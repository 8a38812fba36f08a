// LayerNorm (fwd/bwd, optional fused residual input) and erf-GELU (fwd/bwd), bf16 I/O,
// fp32 statistics.  HBM-bound: one wave per row, 4-byte (bf16x2) lane accesses, the row
// lives in registers between the two statistics passes.
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / 64;

// element pair load/store for the two I/O types (bf16 storage or fp32 storage)
template <typename T> struct IO;
template <> struct IO<bf16_t> {
    static __device__ __forceinline__ void ld2(const bf16_t* p, float& a, float& b) {
        const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
        a = bf2f((bf16_t)(u & 0xffff));
        b = bf2f((bf16_t)(u >> 16));
    }
    static __device__ __forceinline__ void st2(bf16_t* p, float a, float b) {
        *reinterpret_cast<uint32_t*>(p) = pack2bf(a, b);
    }
};
template <> struct IO<float> {
    static __device__ __forceinline__ void ld2(const float* p, float& a, float& b) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        a = v.x;
        b = v.y;
    }
    static __device__ __forceinline__ void st2(float* p, float a, float b) {
        *reinterpret_cast<float2*>(p) = make_float2(a, b);
    }
};

template <int ITERS, typename T>
__device__ __forceinline__ void ln_load_row(float (&xv)[ITERS * 2], const T* x, const T* res, int C, int lane) {
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = i * 128 + lane * 2;
        float a = 0.f, b = 0.f;
        if (c < C) {
            IO<T>::ld2(x + c, a, b);
            if (res) {
                float ra, rb;
                IO<T>::ld2(res + c, ra, rb);
                a += ra;
                b += rb;
            }
        }
        xv[2 * i] = a;
        xv[2 * i + 1] = b;
    }
}

template <int ITERS, typename T>
__global__ void __launch_bounds__(LN_THREADS) ln_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, T* __restrict__ y, T* __restrict__ sum_out, float* __restrict__ mean,
    float* __restrict__ rstd, int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * LN_WAVES;
    const float invC = 1.0f / (float)C;
    float g[ITERS * 2], b[ITERS * 2];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = i * 128 + lane * 2;
        const bool ok = c < C;
        g[2 * i] = ok ? gamma[c] : 0.f;
        g[2 * i + 1] = ok ? gamma[c + 1] : 0.f;
        b[2 * i] = ok ? beta[c] : 0.f;
        b[2 * i + 1] = ok ? beta[c + 1] : 0.f;
    }
    for (int64_t row = wave; row < rows; row += nwaves) {
        float xv[ITERS * 2];
        ln_load_row<ITERS, T>(xv, x + row * C, res ? res + row * C : nullptr, C, lane);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ITERS * 2; ++i) s += xv[i];
        const float mu = wave_sum(s) * invC;
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int c = i * 128 + lane * 2;
            if (c < C) {
                const float d0 = xv[2 * i] - mu, d1 = xv[2 * i + 1] - mu;
                vs += d0 * d0 + d1 * d1;
            }
        }
        const float rs = rsqrtf(wave_sum(vs) * invC + eps);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int c = i * 128 + lane * 2;
            if (c < C) {
                const float o0 = (xv[2 * i] - mu) * rs * g[2 * i] + b[2 * i];
                const float o1 = (xv[2 * i + 1] - mu) * rs * g[2 * i + 1] + b[2 * i + 1];
                IO<T>::st2(y + row * C + c, o0, o1);
                if (sum_out) IO<T>::st2(sum_out + row * C + c, xv[2 * i], xv[2 * i + 1]);
            }
        }
    }
}

template <int ITERS, typename T>
__global__ void __launch_bounds__(LN_THREADS) ln_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    const T* __restrict__ dsum, T* __restrict__ dx, float* __restrict__ partial, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int64_t rows, int C) {
    __shared__ float red[ITERS * 128 * 2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * LN_WAVES + wv;
    const int64_t nwaves = (int64_t)gridDim.x * LN_WAVES;
    const float invC = 1.0f / (float)C;
    float g[ITERS * 2], dg[ITERS * 2], db[ITERS * 2];
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = i * 128 + lane * 2;
        const bool ok = c < C;
        g[2 * i] = ok ? gamma[c] : 0.f;
        g[2 * i + 1] = ok ? gamma[c + 1] : 0.f;
        dg[2 * i] = dg[2 * i + 1] = db[2 * i] = db[2 * i + 1] = 0.f;
    }
    for (int64_t row = wave; row < rows; row += nwaves) {
        float xv[ITERS * 2], dv[ITERS * 2];
        ln_load_row<ITERS, T>(xv, x + row * C, res ? res + row * C : nullptr, C, lane);
        ln_load_row<ITERS, T>(dv, dy + row * C, (const T*)nullptr, C, lane);
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < ITERS * 2; ++i) {
            const int c = (i >> 1) * 128 + lane * 2;
            const float xh = (c < C) ? (xv[i] - mu) * rs : 0.f;
            xv[i] = xh;
            const float gd = g[i] * dv[i];
            s1 += gd;
            s2 += gd * xh;
            dg[i] += dv[i] * xh;
            db[i] += dv[i];
        }
        s1 = wave_sum(s1) * invC;
        s2 = wave_sum(s2) * invC;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int c = i * 128 + lane * 2;
            if (c < C) {
                float o0 = rs * (g[2 * i] * dv[2 * i] - s1 - xv[2 * i] * s2);
                float o1 = rs * (g[2 * i + 1] * dv[2 * i + 1] - s1 - xv[2 * i + 1] * s2);
                if (dsum) {                                  // gradient arriving through the residual path
                    float a0, a1;
                    IO<T>::ld2(dsum + row * C + c, a0, a1);
                    o0 += a0;
                    o1 += a1;
                }
                IO<T>::st2(dx + row * C + c, o0, o1);
            }
        }
    }
    // block reduce dgamma/dbeta over the 4 waves (one shared row, waves take turns), then
    // write one partial row per block
    for (int w = 0; w < LN_WAVES; ++w) {
        if (wv == w) {
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int c = i * 128 + lane * 2;
                if (w == 0) {
                    red[c] = dg[2 * i];
                    red[c + 1] = dg[2 * i + 1];
                    red[ITERS * 128 + c] = db[2 * i];
                    red[ITERS * 128 + c + 1] = db[2 * i + 1];
                } else {
                    red[c] += dg[2 * i];
                    red[c + 1] += dg[2 * i + 1];
                    red[ITERS * 128 + c] += db[2 * i];
                    red[ITERS * 128 + c + 1] += db[2 * i + 1];
                }
            }
        }
        __syncthreads();
    }
    const int nblk = gridDim.x;
    if (dgamma) {                                   // few blocks: add straight into dgamma/dbeta, no second kernel
        for (int c = threadIdx.x; c < C; c += LN_THREADS) {
            atomicAdd(dgamma + c, red[c]);
            atomicAdd(dbeta + c, red[ITERS * 128 + c]);
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += LN_THREADS) {
        partial[(int64_t)blockIdx.x * C + c] = red[c];
        partial[((int64_t)nblk + blockIdx.x) * C + c] = red[ITERS * 128 + c];
    }
}

// dgamma[c] = sum_i partial[0][i][c], dbeta[c] = sum_i partial[1][i][c].  Block = 64 channels x 16
// row-splits (1024 threads): coalesced 256-B row reads, 16-way split of the nblk loop, LDS tree.
__global__ void __launch_bounds__(1024) ln_bwd_reduce_kernel(const float* __restrict__ partial,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int nblk, int C) {
    __shared__ float sa[16][64], sb[16][64];
    const int cl = threadIdx.x & 63, sp = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) {
        for (int i = blockIdx.y * 16 + sp; i < nblk; i += 16 * gridDim.y) {
            a += partial[(int64_t)i * C + c];
            b += partial[((int64_t)nblk + i) * C + c];
        }
    }
    sa[sp][cl] = a;
    sb[sp][cl] = b;
    __syncthreads();
    if (sp == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            a += sa[k][cl];
            b += sb[k][cl];
        }
        atomicAdd(dgamma + c, a);                      // dgamma/dbeta are ACCUMULATED into (caller zeroes)
        atomicAdd(dbeta + c, b);
    }
}

int ln_iters(int C) {
    const int need = (C + 127) / 128;
    const int opts[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    for (int o : opts) if (o >= need) return o;
    return -1;
}

static int env_cap(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
int ln_fwd_blocks(int64_t rows) {
    static const int cap = env_cap("CLV_LN_GRID", 2048);
    int64_t b = (rows + LN_WAVES - 1) / LN_WAVES;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

#define LN_CASE(N, KERNEL, TY, GRID, ...) \
    case N: hipLaunchKernelGGL((KERNEL<N, TY>), dim3(GRID), dim3(LN_THREADS), 0, st, __VA_ARGS__); break;
#define LN_DISPATCH(IT, KERNEL, TY, GRID, ...)                                          \
    switch (IT) {                                                                        \
        LN_CASE(1, KERNEL, TY, GRID, __VA_ARGS__) LN_CASE(2, KERNEL, TY, GRID, __VA_ARGS__) \
        LN_CASE(3, KERNEL, TY, GRID, __VA_ARGS__) LN_CASE(4, KERNEL, TY, GRID, __VA_ARGS__) \
        LN_CASE(6, KERNEL, TY, GRID, __VA_ARGS__) LN_CASE(8, KERNEL, TY, GRID, __VA_ARGS__) \
        LN_CASE(12, KERNEL, TY, GRID, __VA_ARGS__) LN_CASE(16, KERNEL, TY, GRID, __VA_ARGS__) \
        LN_CASE(24, KERNEL, TY, GRID, __VA_ARGS__) LN_CASE(32, KERNEL, TY, GRID, __VA_ARGS__) \
        default: return CLV_ERR_UNSUPPORTED;                                             \
    }

// =====================================================================================================
// Vector LayerNorm (C % 8 == 0, C <= 3072): 16-byte lane accesses; a row is owned by a GROUP of 16 / 32 / 64
// lanes (so C = 96 packs 4 rows per wave, C = 192 two), ITERS chunks of 8 elements per lane.  The x operand
// may carry the transforms that precede the norm in the model, applied on the fly instead of as separate
// elementwise kernels:  t = keep(x) * x / (1 - p) * xscale[sample] + res   (nn.Dropout on the sub-layer
// output, BertSelfOutput / BertOutput; per-sample DropPath scale, swin_transformer_3d.py:498,503).  The
// dropout mask is a pure function of (seed, row, column), so the backward regenerates it.
struct LnX {
    const float* xscale;                 // [rows / rows_per_sample] or null
    int rows_per_sample;
    unsigned thresh;                     // P(drop) = thresh / 2^32, 0 = off
    float inv_keep;
    const unsigned long long* seed;
    int on_load;                         // backward: 1 = x is the raw operand (re-apply), 0 = x already holds t
    // PatchMerging gather (swin_transformer_3d.py:531-539): the row of width C = 4*gC is the concatenation of the
    // gC-wide source rows (2h+hp, 2w+wp) with channel block q = 2*wp + hp; x / res / dx / dres are then the
    // UN-gathered [.., 2*gH2, 2*gW2, gC] tensors and the kernels address them through src_off().  gC = 0: off.
    int gC, gH2, gW2;
    // fp8 path (clv_gemm_nt_fp8's activation operand produced HERE instead of by a clv_quant_fp8_rows pass over y):
    // q8 [rows][C] e4m3 bytes = y / qscale[row], qscale[row] = max |y[row]| / 448.  null: off.
    unsigned char* q8;
    float* qscale;
};

// element offset in the un-gathered tensor of column c (a multiple of 8; gC % 8 == 0) of gathered row `row`
__device__ __forceinline__ int64_t ln_src_off(const LnX& xf, int64_t row, int c, int C) {
    if (!xf.gC) return row * C + c;
    const int q = c / xf.gC;
    const int64_t w2 = row % xf.gW2, t = row / xf.gW2;
    const int64_t h2 = t % xf.gH2, bd = t / xf.gH2;
    const int64_t src = (bd * (2 * xf.gH2) + 2 * h2 + (q & 1)) * (2 * xf.gW2) + 2 * w2 + (q >> 1);
    return src * xf.gC + (c - q * xf.gC);
}

__device__ __forceinline__ float ln_keep(unsigned long long seed, unsigned row, unsigned col, unsigned thresh,
                                         float inv_keep) {
    unsigned x = (row * 0x9E3779B1u) ^ (col * 0x85EBCA77u) ^ (unsigned)seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    x += (unsigned)(seed >> 32);
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
    return (x >= thresh) ? inv_keep : 0.f;
}

template <typename T> struct IO8;
template <> struct IO8<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[8]) {
        Frag8 f;
        f.u4 = *reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(f.h[e]);
    }
    static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[8]) {
        Frag8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) f.u[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<uint4*>(p) = f.u4;
    }
};
template <> struct IO8<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) { IO8<float>::ld(p, v); }

template <int GROUP>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

template <int GROUP>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// t = xform(x) + res for this lane's ITERS chunks of `row`; m[] receives the multiplier applied to x
template <int GROUP, int ITERS, typename T, bool XF>
__device__ __forceinline__ void lnv_load(float (&t)[ITERS][8], float (&m)[ITERS][8], const T* x, const T* res,
                                         int64_t row, int C, int gl, const LnX& xf, bool apply) {
    float xs = 1.f;
    unsigned long long sd = 0;
    if (XF) {
        if (xf.xscale) xs = xf.xscale[row / xf.rows_per_sample];
        if (xf.thresh) sd = *xf.seed;
    }
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
        const int c = (i * GROUP + gl) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) { t[i][e] = 0.f; m[i][e] = 1.f; }
        if (c < C) {
            const int64_t off = XF ? ln_src_off(xf, row, c, C) : row * C + c;
            IO8<T>::ld(x + off, t[i]);
            if (XF) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float k = xs;
                    if (xf.thresh) k *= ln_keep(sd, (unsigned)row, (unsigned)(c + e), xf.thresh, xf.inv_keep);
                    m[i][e] = k;
                    if (apply) t[i][e] *= k;
                }
            }
            if (res) {
                float r[8];
                IO8<T>::ld(res + off, r);
#pragma unroll
                for (int e = 0; e < 8; ++e) t[i][e] += r[e];
            }
        }
    }
}

template <int GROUP, int ITERS, typename T, bool XF>
__global__ void __launch_bounds__(LN_THREADS) lnv_fwd_kernel(
    const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, T* __restrict__ y, T* __restrict__ sum_out, float* __restrict__ mean,
    float* __restrict__ rstd, int64_t rows, int C, float eps, LnX xf) {
    constexpr int RPW = 64 / GROUP;
    const int lane = threadIdx.x & 63, gl = lane & (GROUP - 1), sub = lane / GROUP;
    const int64_t wave = (int64_t)blockIdx.x * LN_WAVES + (threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * LN_WAVES * RPW;
    const float invC = 1.0f / (float)C;
    for (int64_t row0 = wave * RPW; row0 < rows; row0 += stride) {
        const int64_t row = row0 + sub;
        const bool live = row < rows;
        float t[ITERS][8], m[ITERS][8];
        if (live) lnv_load<GROUP, ITERS, T, XF>(t, m, x, res, row, C, gl, xf, true);
        else {
#pragma unroll
            for (int i = 0; i < ITERS; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) t[i][e] = 0.f;
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += t[i][e];
        const float mu = group_sum<GROUP>(s) * invC;
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < ITERS; ++i)
            if ((i * GROUP + gl) * 8 < C) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = t[i][e] - mu;
                    vs += d * d;
                }
            }
        const float rs = rsqrtf(group_sum<GROUP>(vs) * invC + eps);
        if (!live) continue;
        if (gl == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int c = (i * GROUP + gl) * 8;
            if (c < C) {
                float g[8], b[8], o[8];
                ld8f(gamma + c, g);
                ld8f(beta + c, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (t[i][e] - mu) * rs * g[e] + b[e];
                IO8<T>::st(y + row * C + c, o);
                if (sum_out) IO8<T>::st(sum_out + row * C + c, t[i]);
                if (xf.q8) {                                  // keep the outputs for the quantisation pass below
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[i][e] = o[e];
                }
            }
        }
        if (xf.q8) {
            float amax = 0.f;
#pragma unroll
            for (int i = 0; i < ITERS; ++i)
                if ((i * GROUP + gl) * 8 < C) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(t[i][e]));
                }
            amax = group_max<GROUP>(amax);
            const float inv = amax > 0.f ? 448.0f / amax : 1.0f;
            if (gl == 0) xf.qscale[row] = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int c = (i * GROUP + gl) * 8;
                if (c < C) {
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(t[i][e] * inv, -448.f), 448.f);
                    int lo = 0, hi = 0;
                    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
                    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
                    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
                    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
                    *reinterpret_cast<uint2*>(xf.q8 + row * C + c) = make_uint2((unsigned)lo, (unsigned)hi);
                }
            }
        }
    }
}

#ifndef LNV_BWD_RG2
#define LNV_BWD_RG2 3       // row groups in flight per trip (one-chunk-per-lane bf16 kernels); 1 = the plain loop
#endif
#ifndef LNV_BWD_WAVES
#define LNV_BWD_WAVES (LNV_BWD_RG2 >= 3 ? 3 : LNV_BWD_RG2 == 2 ? 4 : 5)
#endif
// ... for rows of <= 32 chunks (C <= 256: Swin stages 0-1, 50 000-200 000 rows = 3-5 full trips per wave).  One row per wave
// (C = 384, 12 544 rows: 4 rows per wave) stays on the plain loop at 5 waves: three-at-a-time leaves it a ragged second trip
// and the transform variants spill there (19.6 vs 17 us per launch).
#define LNV_RG_ON(GROUP, ITERS, T) (LNV_BWD_RG2 > 1 && (ITERS) == 1 && sizeof(T) == 2 && (GROUP) <= 32)
#define LNV_BWD_MINW(GROUP, ITERS, T) ((ITERS) != 1 ? 1 : (LNV_RG_ON(GROUP, ITERS, T) ? LNV_BWD_WAVES : 5))
// dx = d t * (x multiplier) [-> dx], d t itself [-> dres, when given];  d t = LN-backward(dy (+ dy2)) (+ dsum)
template <int GROUP, int ITERS, typename T, bool XF>
// one 16-byte chunk per lane (ITERS == 1): 96 VGPRs = 5 waves per SIMD (the transform variants wanted 98-112: 4)
__global__ void __launch_bounds__(LN_THREADS, LNV_BWD_MINW(GROUP, ITERS, T)) lnv_bwd_kernel(
    const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ x, const T* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
    const T* __restrict__ dsum, T* __restrict__ dx, T* __restrict__ dres, float* __restrict__ partial,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int C, LnX xf) {
    constexpr int RPW = 64 / GROUP;
    constexpr int CW = ITERS * GROUP * 8;               // padded row width
    __shared__ float red[2 * CW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gl = lane & (GROUP - 1), sub = lane / GROUP;
    const int64_t wave = (int64_t)blockIdx.x * LN_WAVES + wv;
    const int64_t stride = (int64_t)gridDim.x * LN_WAVES * RPW;
    const float invC = 1.0f / (float)C;
    float dg[ITERS][8], db[ITERS][8];
#pragma unroll
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dg[i][e] = db[i][e] = 0.f;
    // One chunk per lane, bf16 storage (Swin stages 0-2: C = 96 / 192 / 384): RG row groups per trip, the raw 16-byte loads of
    // all of them issued before anything is converted — a wave keeps RG times the bytes in flight.  These kernels run at the
    // rate their waves' loads are outstanding, not at the HBM's (the plain loop at 4 instead of 5 waves per SIMD: +25 %,
    // 11.24 -> 11.37 ms per step).  The groups are then worked off one at a time.  Same box, ms per step: plain loop, 5 waves
    // 11.41; RG = 2, 4 waves 11.45; RG = 3, 3 waves (<= 168 VGPRs) 11.33.
    if (LNV_RG_ON(GROUP, ITERS, T) && rows * C < (1ll << 31)) {      // 32-bit element offsets below
        constexpr int RG = LNV_BWD_RG2 > 1 ? LNV_BWD_RG2 : 2;
        const int c = gl * 8;
        const bool cv = c < C;
        float gm[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) gm[e] = 0.f;
        if (cv) ld8f(gamma + c, gm);
        for (int64_t row0 = wave * RPW; row0 < rows; row0 += RG * stride) {
            int64_t row[RG];
            unsigned off[RG], offd[RG];
            bool live[RG];
            uint4 rx[RG], rr[RG], rd[RG], rd2[RG], ra[RG];
            float mu[RG], rs[RG];
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                row[g] = row0 + g * stride + sub;
                live[g] = row[g] < rows && cv;
                rx[g] = rr[g] = rd[g] = rd2[g] = ra[g] = make_uint4(0, 0, 0, 0);
                mu[g] = rs[g] = 0.f;
                off[g] = offd[g] = 0;
                if (live[g]) {
                    // wave-uniform bases + 32-bit lane offsets: a 64-bit address pair per load is what does not fit
                    offd[g] = (unsigned)(row[g] * C + c);
                    off[g] = XF ? (unsigned)ln_src_off(xf, row[g], c, C) : offd[g];
                    rx[g] = *reinterpret_cast<const uint4*>(x + off[g]);
                    if (res) rr[g] = *reinterpret_cast<const uint4*>(res + off[g]);
                    rd[g] = *reinterpret_cast<const uint4*>(dy + offd[g]);
                    if (dy2) rd2[g] = *reinterpret_cast<const uint4*>(dy2 + offd[g]);
                    if (dsum) ra[g] = *reinterpret_cast<const uint4*>(dsum + offd[g]);
                    mu[g] = mean[row[g]];
                    rs[g] = rstd[row[g]];
                }
            }
#pragma unroll
            for (int g = 0; g < RG; ++g) {
                auto lo = [](uint32_t w) { return half_lo(w); };
                auto hi = [](uint32_t w) { return half_hi(w); };
                const uint32_t* wx = reinterpret_cast<const uint32_t*>(&rx[g]);      // (register views: these are values)
                float tt[8], dd[8];
                const uint32_t ux[4] = {rx[g].x, rx[g].y, rx[g].z, rx[g].w}, ur[4] = {rr[g].x, rr[g].y, rr[g].z, rr[g].w};
                const uint32_t ud[4] = {rd[g].x, rd[g].y, rd[g].z, rd[g].w}, ud2[4] = {rd2[g].x, rd2[g].y, rd2[g].z, rd2[g].w};
                const uint32_t ua[4] = {ra[g].x, ra[g].y, ra[g].z, ra[g].w};
                (void)wx;
                float xs = 1.f;
                unsigned long long sd = 0;
                if (XF && live[g]) {
                    if (xf.xscale) xs = xf.xscale[row[g] / xf.rows_per_sample];
                    if (xf.thresh) sd = *xf.seed;
                }
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xv = (e & 1) ? hi(ux[e >> 1]) : lo(ux[e >> 1]);
                    const float rv = (e & 1) ? hi(ur[e >> 1]) : lo(ur[e >> 1]);
                    float k = 1.f;
                    if (XF) {
                        k = xs;
                        if (xf.thresh) k *= ln_keep(sd, (unsigned)row[g], (unsigned)(c + e), xf.thresh, xf.inv_keep);
                    }
                    const float tv = ((XF && xf.on_load != 0) ? xv * k : xv) + rv;
                    const float dv = ((e & 1) ? hi(ud[e >> 1]) : lo(ud[e >> 1])) + ((e & 1) ? hi(ud2[e >> 1]) : lo(ud2[e >> 1]));
                    const float xh = live[g] ? (tv - mu[g]) * rs[g] : 0.f;
                    tt[e] = xh;
                    dg[0][e] += dv * xh;
                    db[0][e] += dv;
                    dd[e] = dv * gm[e];
                    s1 += dd[e];
                    s2 += dd[e] * xh;
                }
                s1 = group_sum<GROUP>(s1) * invC;
                s2 = group_sum<GROUP>(s2) * invC;
                if (live[g]) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        o[e] = rs[g] * (dd[e] - s1 - tt[e] * s2) + ((e & 1) ? hi(ua[e >> 1]) : lo(ua[e >> 1]));
                    if (dres) IO8<T>::st(dres + off[g], o);
                    if (XF) {                                // the x multiplier again (DropPath scale; dropout keep re-hashed)
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float k = xs;
                            if (xf.thresh) k *= ln_keep(sd, (unsigned)row[g], (unsigned)(c + e), xf.thresh, xf.inv_keep);
                            o[e] *= k;
                        }
                    }
                    IO8<T>::st(dx + off[g], o);
                }
                __builtin_amdgcn_sched_barrier(0);           // one group's fp32 expansion at a time
            }
        }
    } else
    for (int64_t row0 = wave * RPW; row0 < rows; row0 += stride) {
        const int64_t row = row0 + sub;
        const bool live = row < rows;
        float t[ITERS][8], m[ITERS][8], d[ITERS][8];
        float mu = 0.f, rs = 0.f;
        if (live) {
            lnv_load<GROUP, ITERS, T, XF>(t, m, x, res, row, C, gl, xf, xf.on_load != 0);
            mu = mean[row];
            rs = rstd[row];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int c = (i * GROUP + gl) * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) d[i][e] = 0.f;
            if (live && c < C) {
                IO8<T>::ld(dy + row * C + c, d[i]);
                if (dy2) {
                    float d2[8];
                    IO8<T>::ld(dy2 + row * C + c, d2);
#pragma unroll
                    for (int e = 0; e < 8; ++e) d[i][e] += d2[e];
                }
                float g[8];
                ld8f(gamma + c, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = (t[i][e] - mu) * rs;
                    t[i][e] = xh;
                    dg[i][e] += d[i][e] * xh;
                    db[i][e] += d[i][e];
                    d[i][e] *= g[e];                    // g * dy
                    s1 += d[i][e];
                    s2 += d[i][e] * xh;
                }
            }
        }
        s1 = group_sum<GROUP>(s1) * invC;
        s2 = group_sum<GROUP>(s2) * invC;
        if (!live) continue;
#pragma unroll
        for (int i = 0; i < ITERS; ++i) {
            const int c = (i * GROUP + gl) * 8;
            if (c < C) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = rs * (d[i][e] - s1 - t[i][e] * s2);
                if (dsum) {                              // gradient arriving through the residual stream
                    float a[8];
                    IO8<T>::ld(dsum + row * C + c, a);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += a[e];
                }
                const int64_t off = XF ? ln_src_off(xf, row, c, C) : row * C + c;
                if (dres) IO8<T>::st(dres + off, o);
                if (XF) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] *= m[i][e];
                }
                IO8<T>::st(dx + off, o);
            }
        }
    }
    // dgamma/dbeta: fold the row groups of the wave, then the 4 waves through LDS (they take turns)
#pragma unroll
    for (int i = 0; i < ITERS; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = GROUP; o < 64; o <<= 1) {
                dg[i][e] += __shfl_xor(dg[i][e], o, 64);
                db[i][e] += __shfl_xor(db[i][e], o, 64);
            }
        }
    for (int w = 0; w < LN_WAVES; ++w) {
        if (wv == w && sub == 0) {
#pragma unroll
            for (int i = 0; i < ITERS; ++i) {
                const int c = (i * GROUP + gl) * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if (w == 0) {
                        red[c + e] = dg[i][e];
                        red[CW + c + e] = db[i][e];
                    } else {
                        red[c + e] += dg[i][e];
                        red[CW + c + e] += db[i][e];
                    }
                }
            }
        }
        __syncthreads();
    }
    const int nblk = gridDim.x;
    if (dgamma) {                                   // few blocks: add straight into dgamma/dbeta, no second kernel
        for (int c = threadIdx.x; c < C; c += LN_THREADS) {
            atomicAdd(dgamma + c, red[c]);
            atomicAdd(dbeta + c, red[CW + c]);
        }
        return;
    }
    for (int c = threadIdx.x; c < C; c += LN_THREADS) {
        partial[(int64_t)blockIdx.x * C + c] = red[c];
        partial[((int64_t)nblk + blockIdx.x) * C + c] = red[CW + c];
    }
}

struct LnvCfg { int group, iters; };
inline bool lnv_config(int C, LnvCfg& cfg) {
    if (C % 8 || C > 3072) return false;
    const int chunks = C / 8;
    if (chunks <= 16) cfg = {16, 1};
    else if (chunks <= 32) cfg = {32, 1};
    else {
        const int need = (chunks + 63) / 64;
        const int opts[] = {1, 2, 3, 4, 6};
        cfg.group = 64;
        cfg.iters = 6;
        for (int o : opts) if (o >= need) { cfg.iters = o; break; }
    }
    return true;
}
inline int lnv_blocks(int64_t rows, int group, bool fwd = false, int iters = 1) {
    const int rpb = LN_WAVES * (64 / group);           // rows per block per pass
    // forward: one pass per wave (-0.08 ms); backward: bounded (dgamma / dbeta partials) — at 1 280 = 256 CUs x the 5 workgroups
    // a CU holds (one 16-byte chunk per lane: 96 VGPRs, LNV_BWD_WAVES), so that every workgroup is resident and walks the same
    // number of rows: 2 048 ran as one full round plus a 60 % one (round 4, same box: 11.99 -> 11.84 ms per step; 1 024, 1 536,
    // 2 560: 11.98, 12.14, 12.08)
    // (since the one-chunk kernels keep three row groups in flight at 3 waves per SIMD: 768 resident workgroups)
    static const int cap_b = env_cap("CLV_LNV_GRID", 0), cap_f = env_cap("CLV_LNV_FWD_GRID", 1 << 20);
    // rows wider than 512 elements (two+ chunks per lane: C = 768 of stage 3 / the fusion encoder, 3 136-3 648 rows): 512
    // blocks of ~2 rows per wave write half the dgamma / dbeta partial rows of 912 one-row-per-wave blocks (5.6 MB beside 22 MB
    // of operands): 11.39 -> 11.36 ms per step, twice; 256: 11.47
    static const int cap_b2 = env_cap("CLV_LNV_GRID2", 512);
    // one chunk per lane: every workgroup resident — 3 per CU where three row groups are in flight, else 5
    const int cap = fwd ? cap_f : (iters > 1 ? cap_b2 : (cap_b > 0 ? cap_b : (group <= 32 && LNV_BWD_RG2 > 1 ? LNV_BWD_WAVES : 5) * 256));
    int64_t b = (rows + rpb - 1) / rpb;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

#define LNV_CASE(G, I, KERNEL, TY, XF, GRID, ...)                                                             \
    if (cfg.group == G && cfg.iters == I) {                                                                   \
        if (XF) hipLaunchKernelGGL((KERNEL<G, I, TY, true>), dim3(GRID), dim3(LN_THREADS), 0, st, __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<G, I, TY, false>), dim3(GRID), dim3(LN_THREADS), 0, st, __VA_ARGS__);   \
    }
#define LNV_DISPATCH(KERNEL, TY, XF, GRID, ...)                                                 \
    LNV_CASE(16, 1, KERNEL, TY, XF, GRID, __VA_ARGS__) LNV_CASE(32, 1, KERNEL, TY, XF, GRID, __VA_ARGS__) \
    LNV_CASE(64, 1, KERNEL, TY, XF, GRID, __VA_ARGS__) LNV_CASE(64, 2, KERNEL, TY, XF, GRID, __VA_ARGS__) \
    LNV_CASE(64, 3, KERNEL, TY, XF, GRID, __VA_ARGS__) LNV_CASE(64, 4, KERNEL, TY, XF, GRID, __VA_ARGS__) \
    LNV_CASE(64, 6, KERNEL, TY, XF, GRID, __VA_ARGS__)

inline bool make_lnx(const ClvLnExtra* ex, LnX& xf) {
    xf = LnX{nullptr, 1, 0u, 1.f, nullptr, 0, 0, 0, 0, nullptr, nullptr};
    if (!ex) return false;
    if (ex->q8 && ex->qscale) {
        xf.q8 = (unsigned char*)ex->q8;
        xf.qscale = ex->qscale;
    }
    if (ex->gather_c > 0) {
        xf.gC = ex->gather_c;
        xf.gH2 = ex->gather_h2;
        xf.gW2 = ex->gather_w2;
    }
    xf.xscale = ex->xscale;
    xf.rows_per_sample = ex->rows_per_sample > 0 ? ex->rows_per_sample : 1;
    if (ex->drop_p > 0.f) {
        xf.thresh = (unsigned)((double)ex->drop_p * 4294967296.0);
        xf.inv_keep = 1.0f / (1.0f - ex->drop_p);
        xf.seed = (const unsigned long long*)ex->seed;
    }
    xf.on_load = ex->x_is_sum ? 0 : 1;
    return xf.xscale != nullptr || xf.thresh != 0 || xf.gC != 0;
}

// --------------------------------------------------------------------------- GELU
__global__ void gelu_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int64_t n8, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        Frag8 a, o;
        a.u4 = *reinterpret_cast<const uint4*>(x + i * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o.u[e] = pack2bf(gelu_erf(bf2f(a.h[2 * e])), gelu_erf(bf2f(a.h[2 * e + 1])));
        *reinterpret_cast<uint4*>(y + i * 8) = o.u4;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
        const int64_t i = n8 * 8 + threadIdx.x;
        y[i] = f2bf(gelu_erf(bf2f(x[i])));
    }
}

__global__ void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                bf16_t* __restrict__ dx, int64_t n8, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        Frag8 a, d, o;
        a.u4 = *reinterpret_cast<const uint4*>(x + i * 8);
        d.u4 = *reinterpret_cast<const uint4*>(dy + i * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o.u[e] = pack2bf(bf2f(d.h[2 * e]) * gelu_erf_grad(bf2f(a.h[2 * e])),
                             bf2f(d.h[2 * e + 1]) * gelu_erf_grad(bf2f(a.h[2 * e + 1])));
        *reinterpret_cast<uint4*>(dx + i * 8) = o.u4;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n8 * 8)) {
        const int64_t i = n8 * 8 + threadIdx.x;
        dx[i] = f2bf(bf2f(dy[i]) * gelu_erf_grad(bf2f(x[i])));
    }
}

__global__ void gelu_fwd_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = gelu_erf(x[i]);
}
__global__ void gelu_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    float* __restrict__ dx, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dx[i] = dy[i] * gelu_erf_grad(x[i]);
}

int ew_blocks(int64_t n8) {
    static const int cap = env_cap("CLV_EW_GRID", 1 << 20);   // one 16-byte group per thread: -0.14 ms per step vs 2048 grid-stride blocks
    int64_t b = (n8 + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int clv_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                                 void* y, void* sum_out, float* mean, float* rstd, int64_t rows, int32_t C,
                                 float eps, int32_t is_f32, const ClvLnExtra* extra, void* stream) {
    if (!x || !gamma || !beta || !y || rows < 0 || C <= 0 || (C & 1)) return CLV_ERR_ARG;
    if (rows == 0) return CLV_OK;
    hipStream_t st = (hipStream_t)stream;
    LnX xf;
    const bool XF = make_lnx(extra, xf);
    if (xf.thresh && !xf.seed) return CLV_ERR_ARG;
    if (xf.gC && (xf.gC * 4 != C || (xf.gC & 7) || xf.gH2 <= 0 || xf.gW2 <= 0 || rows % ((int64_t)xf.gH2 * xf.gW2) ||
                  sum_out || xf.thresh))
        return CLV_ERR_ARG;
    LnvCfg cfg;
    if (lnv_config(C, cfg)) {
        const int grid = lnv_blocks(rows, cfg.group, true);
        if (is_f32) {
            LNV_DISPATCH(lnv_fwd_kernel, float, XF, grid, (const float*)x, (const float*)res, gamma, beta, (float*)y,
                         (float*)sum_out, mean, rstd, rows, (int)C, eps, xf)
        } else {
            LNV_DISPATCH(lnv_fwd_kernel, bf16_t, XF, grid, (const bf16_t*)x, (const bf16_t*)res, gamma, beta,
                         (bf16_t*)y, (bf16_t*)sum_out, mean, rstd, rows, (int)C, eps, xf)
        }
        return clv_check_launch();
    }
    if (XF || xf.q8) return CLV_ERR_UNSUPPORTED;     // the scalar fallback has no operand transforms / fp8 output
    const int it = ln_iters(C);
    const int grid = ln_fwd_blocks(rows);
    if (is_f32) {
        LN_DISPATCH(it, ln_fwd_kernel, float, grid, (const float*)x, (const float*)res, gamma, beta, (float*)y,
                    (float*)sum_out, mean, rstd, rows, (int)C, eps)
    } else {
        LN_DISPATCH(it, ln_fwd_kernel, bf16_t, grid, (const bf16_t*)x, (const bf16_t*)res, gamma, beta,
                    (bf16_t*)y, (bf16_t*)sum_out, mean, rstd, rows, (int)C, eps)
    }
    return clv_check_launch();
}

extern "C" int clv_layernorm_bwd_blocks(int64_t rows, int32_t C) {
    LnvCfg cfg;
    if (lnv_config(C, cfg)) return lnv_blocks(rows, cfg.group, false, cfg.iters);
    int64_t b = (rows + LN_WAVES - 1) / LN_WAVES;      // one row per wave until the chip is full
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

extern "C" int clv_layernorm_bwd(const void* dy, const void* x, const void* res, const float* gamma,
                                 const float* mean, const float* rstd, const void* dsum, void* dx, float* dgamma,
                                 float* dbeta, float* partial, int64_t rows, int32_t C, int32_t is_f32,
                                 const ClvLnExtra* extra, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !partial || rows <= 0 || C <= 0 ||
        (C & 1))
        return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    LnX xf;
    const bool XF = make_lnx(extra, xf);
    if (xf.thresh && !xf.seed) return CLV_ERR_ARG;
    if (xf.gC && (xf.gC * 4 != C || (xf.gC & 7) || xf.gH2 <= 0 || xf.gW2 <= 0 || rows % ((int64_t)xf.gH2 * xf.gW2) ||
                  dsum || xf.thresh || !xf.on_load))
        return CLV_ERR_ARG;
    const void* dy2 = extra ? extra->dy2 : nullptr;
    void* dres = extra ? extra->dres : nullptr;
    const int grid = clv_layernorm_bwd_blocks(rows, C);
    const bool direct = grid <= 256;              // up to 256 partial rows: atomics beat a second launch
    float* dgd = direct ? dgamma : nullptr;
    float* dbd = direct ? dbeta : nullptr;
    LnvCfg cfg;
    if (lnv_config(C, cfg)) {
        if (is_f32) {
            LNV_DISPATCH(lnv_bwd_kernel, float, XF, grid, (const float*)dy, (const float*)dy2, (const float*)x,
                         (const float*)res, gamma, mean, rstd, (const float*)dsum, (float*)dx, (float*)dres, partial,
                         dgd, dbd, rows, (int)C, xf)
        } else {
            LNV_DISPATCH(lnv_bwd_kernel, bf16_t, XF, grid, (const bf16_t*)dy, (const bf16_t*)dy2, (const bf16_t*)x,
                         (const bf16_t*)res, gamma, mean, rstd, (const bf16_t*)dsum, (bf16_t*)dx, (bf16_t*)dres,
                         partial, dgd, dbd, rows, (int)C, xf)
        }
    } else {
        if (XF || dy2 || dres) return CLV_ERR_UNSUPPORTED;
        const int it = ln_iters(C);
        if (is_f32) {
            LN_DISPATCH(it, ln_bwd_kernel, float, grid, (const float*)dy, (const float*)x, (const float*)res, gamma,
                        mean, rstd, (const float*)dsum, (float*)dx, partial, dgd, dbd, rows, (int)C)
        } else {
            LN_DISPATCH(it, ln_bwd_kernel, bf16_t, grid, (const bf16_t*)dy, (const bf16_t*)x, (const bf16_t*)res,
                        gamma, mean, rstd, (const bf16_t*)dsum, (bf16_t*)dx, partial, dgd, dbd, rows, (int)C)
        }
    }
    int rc = clv_check_launch();
    if (rc || direct || (extra && extra->no_reduce)) return rc;
    const int ysplit = grid >= 256 ? 8 : 1;      // > 1: atomics into dgamma/dbeta (caller zeroes them)
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((C + 63) / 64, ysplit), dim3(1024), 0, st, partial, dgamma, dbeta,
                       grid, (int)C);
    return clv_check_launch();
}

extern "C" int clv_layernorm_bwd_needs_reduce(int64_t rows, int32_t C) { return clv_layernorm_bwd_blocks(rows, C) > 256; }

namespace {
struct LnReduceTable {
    ClvLnReduceEntry e[CLV_LN_REDUCE_MAX];
    int n;
};
// block = 64 channels x 16 row-lanes of one entry (see ln_bwd_reduce_kernel); LNR_YS block rows per (entry, channel
// group): 16 instead of the first version's 4 — 8 instead of 32 dependent iterations per thread
constexpr int LNR_YS = 16;
__global__ void __launch_bounds__(1024) ln_reduce_batch_kernel(LnReduceTable tab) {
    __shared__ float sa[16][64], sb[16][64];
    int idx = 0;
    for (int i = 1; i < tab.n; ++i)
        if ((int)blockIdx.x >= tab.e[i].block_begin) idx = i;
    const ClvLnReduceEntry& en = tab.e[idx];
    const int lb = blockIdx.x - en.block_begin, cb = lb / LNR_YS, ys = lb % LNR_YS;
    const int cl = threadIdx.x & 63, sp = threadIdx.x >> 6;
    const int c = cb * 64 + cl, C = en.C, nblk = en.nblk;
    float a = 0.f, b = 0.f;
    if (c < C) {
        for (int i = ys * 16 + sp; i < nblk; i += 16 * LNR_YS) {
            a += en.partial[(int64_t)i * C + c];
            b += en.partial[((int64_t)nblk + i) * C + c];
        }
    }
    sa[sp][cl] = a;
    sb[sp][cl] = b;
    __syncthreads();
    if (sp == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            a += sa[k][cl];
            b += sb[k][cl];
        }
        atomicAdd(en.dgamma + c, a);
        atomicAdd(en.dbeta + c, b);
    }
}
}  // namespace

extern "C" int clv_ln_reduce_batch(const ClvLnReduceEntry* entries, int32_t n, void* stream) {
    if (!entries || n <= 0 || n > CLV_LN_REDUCE_MAX) return CLV_ERR_ARG;
    static_assert(sizeof(ClvLnReduceEntry) == 40, "ClvLnReduceEntry layout is part of the ABI");
    LnReduceTable tab;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        ClvLnReduceEntry en = entries[i];
        if (!en.partial || !en.dgamma || !en.dbeta || en.nblk <= 0 || en.C <= 0) return CLV_ERR_ARG;
        en.block_begin = blocks;
        blocks += LNR_YS * ((en.C + 63) / 64);
        tab.e[i] = en;
    }
    tab.n = n;
    hipLaunchKernelGGL(ln_reduce_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, tab);
    return clv_check_launch();
}

extern "C" int clv_gelu_fwd(const void* x, void* y, int64_t n, int32_t is_f32, void* stream) {
    if (!x || !y || n < 0) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if (is_f32) {
        hipLaunchKernelGGL(gelu_fwd_f32_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)x, (float*)y, n);
        return clv_check_launch();
    }
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(ew_blocks(n8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)y, n8, n);
    return clv_check_launch();
}

extern "C" int clv_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int32_t is_f32, void* stream) {
    if (!dy || !x || !dx || n < 0) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if (is_f32) {
        hipLaunchKernelGGL(gelu_bwd_f32_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)dy, (const float*)x, (float*)dx, n);
        return clv_check_launch();
    }
    const int64_t n8 = n / 8;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(ew_blocks(n8)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, n8, n);
    return clv_check_launch();
}

// ------------------------------------------------------------------ LayerNorm affine folded into the next Linear
// (xhat * gamma + beta) W^T + b  ==  xhat (W * gamma)^T + (b + W beta): the fused LN+projection kernels take the
// standardised rows, so gamma / beta live in the weights.  One small kernel each way instead of ~16 elementwise /
// mv / reduction launches of autograd-tracked torch ops per layer and step.
namespace {

// block = one output row n: wf[n][:] = bf16(w[n][:] * gamma), bf[n] = b[n] + sum_k w[n][k] beta[k]
__global__ void __launch_bounds__(128) ln_fold_fwd_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          bf16_t* __restrict__ wf, float* __restrict__ bf, int K) {
    __shared__ float red[2];
    const int n = blockIdx.x;
    float dot = 0.f;
    for (int k = threadIdx.x; k < K; k += 128) {
        const float v = w[(int64_t)n * K + k];
        wf[(int64_t)n * K + k] = f2bf(v * gamma[k]);
        dot += v * beta[k];
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) bf[n] = (b ? b[n] : 0.f) + red[0] + red[1];
}

// One pass over (dwf, w), ROWS rows per block, a thread per column:
//   dW[n][k] += dwf[n][k] gamma[k] + dbf[n] beta[k];  db[n] += dbf[n];
//   dgamma[k] += sum_n dwf[n][k] w[n][k];  dbeta[k] += sum_n dbf[n] w[n][k]   (block partials -> one atomic per column)
// (was two kernels, the column sums a 2-block launch that walked all N rows serially: 30-40 us at N = 384.)
constexpr int LNF_ROWS = 4;
__global__ void __launch_bounds__(128) ln_fold_bwd_kernel(const float* __restrict__ dwf, const float* __restrict__ dbf,
                                                          const float* __restrict__ w, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ dw,
                                                          float* __restrict__ db, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int N, int K) {
    const int n0 = blockIdx.x * LNF_ROWS;
    float dbn[LNF_ROWS];
#pragma unroll
    for (int r = 0; r < LNF_ROWS; ++r) dbn[r] = n0 + r < N ? dbf[n0 + r] : 0.f;
    for (int k = threadIdx.x; k < K; k += 128) {
        const float gk = gamma[k], bk = beta[k];
        float g = 0.f, bb = 0.f;
#pragma unroll
        for (int r = 0; r < LNF_ROWS; ++r) {
            if (n0 + r >= N) break;
            const int64_t e = (int64_t)(n0 + r) * K + k;
            const float wv = w[e], dv = dwf[e];
            dw[e] += dv * gk + dbn[r] * bk;
            g += dv * wv;
            bb += dbn[r] * wv;
        }
        atomicAdd(dgamma + k, g);
        atomicAdd(dbeta + k, bb);
    }
    if (db && threadIdx.x < LNF_ROWS && n0 + threadIdx.x < N) db[n0 + threadIdx.x] += dbf[n0 + threadIdx.x];
}


// ---------------------------------------------------------------------------------------------------
// nn.BatchNorm1d over [B][D] fp32 — the ln=False / text_bn=True variants of the contrastive projection heads
// (ssl_head.py:50-66,175-186,252-262): B = clips of one rank (8 .. a few hundred), D = 768 .. 1536.  One thread owns a
// column and walks the rows (adjacent threads read adjacent columns: coalesced); two-pass variance as torch computes it.
__global__ void __launch_bounds__(256) bn1d_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ y,
                                                       float* __restrict__ save_mean, float* __restrict__ save_rstd, int B,
                                                       int D, float eps, float momentum, int training) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= D) return;
    float mean, rstd;
    if (training) {
        float s = 0.f;
        for (int b = 0; b < B; ++b) s += x[(int64_t)b * D + j];
        mean = s / (float)B;
        float v = 0.f;
        for (int b = 0; b < B; ++b) {
            const float d = x[(int64_t)b * D + j] - mean;
            v += d * d;
        }
        const float var = v / (float)B;                      // biased: what normalises the batch
        rstd = rsqrtf(var + eps);
        if (running_mean) {                                  // running statistics take the UNBIASED variance (torch)
            running_mean[j] = (1.f - momentum) * running_mean[j] + momentum * mean;
            running_var[j] = (1.f - momentum) * running_var[j] + momentum * (B > 1 ? v / (float)(B - 1) : var);
        }
    } else {
        mean = running_mean[j];
        rstd = rsqrtf(running_var[j] + eps);
    }
    save_mean[j] = mean;
    save_rstd[j] = rstd;
    const float g = gamma ? gamma[j] : 1.f, bt = beta ? beta[j] : 0.f;
    for (int b = 0; b < B; ++b) y[(int64_t)b * D + j] = (x[(int64_t)b * D + j] - mean) * rstd * g + bt;
}

__global__ void __launch_bounds__(256) bn1d_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                       const float* __restrict__ gamma, const float* __restrict__ save_mean,
                                                       const float* __restrict__ save_rstd, float* __restrict__ dx,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int D,
                                                       int training) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= D) return;
    const float mean = save_mean[j], rstd = save_rstd[j], g = gamma ? gamma[j] : 1.f;
    float sdy = 0.f, sdyx = 0.f;
    for (int b = 0; b < B; ++b) {
        const float d = dy[(int64_t)b * D + j];
        sdy += d;
        sdyx += d * (x[(int64_t)b * D + j] - mean) * rstd;
    }
    if (dgamma) dgamma[j] = sdyx;
    if (dbeta) dbeta[j] = sdy;
    if (!dx) return;
    const float inv = 1.f / (float)B;
    for (int b = 0; b < B; ++b) {
        const float d = dy[(int64_t)b * D + j];
        // eval mode: the statistics are constants; training mode: they depend on every row of the column
        dx[(int64_t)b * D + j] = training ? g * rstd * (d - inv * sdy - (x[(int64_t)b * D + j] - mean) * rstd * inv * sdyx) : g * rstd * d;
    }
}

}  // namespace

extern "C" int clv_ln_fold_fwd(const float* w, const float* b, const float* gamma, const float* beta, void* wf,
                               float* bf, int32_t N, int32_t K, void* stream) {
    if (!w || !gamma || !beta || !wf || !bf || N <= 0 || K <= 0) return CLV_ERR_ARG;
    hipLaunchKernelGGL(ln_fold_fwd_kernel, dim3(N), dim3(128), 0, (hipStream_t)stream, w, b, gamma, beta, (bf16_t*)wf, bf,
                       (int)K);
    return clv_check_launch();
}

extern "C" int clv_ln_fold_bwd(const float* dwf, const float* dbf, const float* w, const float* gamma,
                               const float* beta, float* dw, float* db, float* dgamma, float* dbeta, int32_t N,
                               int32_t K, void* stream) {
    if (!dwf || !dbf || !w || !gamma || !beta || !dw || !dgamma || !dbeta || N <= 0 || K <= 0) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ln_fold_bwd_kernel, dim3((N + LNF_ROWS - 1) / LNF_ROWS), dim3(128), 0, st, dwf, dbf, w, gamma, beta, dw,
                       db, dgamma, dbeta, (int)N, (int)K);
    return clv_check_launch();
}

extern "C" int clv_batchnorm1d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* y, float* save_mean, float* save_rstd, int32_t B, int32_t D,
                                   float eps, float momentum, int32_t training, void* stream) {
    if (!x || !y || !save_mean || !save_rstd || B <= 0 || D <= 0) return CLV_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return CLV_ERR_ARG;
    if (!training && !running_mean) return CLV_ERR_ARG;
    hipLaunchKernelGGL(bn1d_fwd_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, running_mean,
                       running_var, y, save_mean, save_rstd, (int)B, (int)D, eps, momentum, (int)training);
    return clv_check_launch();
}

extern "C" int clv_batchnorm1d_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean,
                                   const float* save_rstd, float* dx, float* dgamma, float* dbeta, int32_t B, int32_t D,
                                   int32_t training, void* stream) {
    if (!dy || !x || !save_mean || !save_rstd || B <= 0 || D <= 0) return CLV_ERR_ARG;
    hipLaunchKernelGGL(bn1d_bwd_kernel, dim3((D + 255) / 256), dim3(256), 0, (hipStream_t)stream, dy, x, gamma, save_mean,
                       save_rstd, dx, dgamma, dbeta, (int)B, (int)D, (int)training);
    return clv_check_launch();
}
