// Shared device helpers for the Clover HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CLV_OK 0
#define CLV_ERR_ARG (-1)
#define CLV_ERR_UNSUPPORTED (-2)
#define CLV_ERR_LAUNCH (-3)

// ---- the 16-bit element type of activations, weight shadows and gradients.  Default: bf16.  With -DCLV_HALF_F16 the SAME
// kernels are compiled for IEEE fp16 (libclover_hip_f16.so, CLOVER_HALF=f16): the reference's own arithmetic type
// (configs/exp_local/pretrain_webvid_cc3m.py:21 fp16 = dict(loss_scale='dynamic')) — three more significand bits at the same
// MFMA rate (v_mfma_f32_16x16x32_f16), which is what the contrastive losses (cosines / 0.05) need to sit within ~1e-3 of the
// fp32 reference (tools/f16_forward_study.py).  Everything type-specific is in this block: the raw 16-bit storage type keeps
// its name (bf16_t), and so do the helpers (bf2f, pack2bf, f2bf, mfma16) — in the f16 build they are the f16 conversions.
typedef uint16_t bf16_t;  // raw 16-bit element (bf16 bits, or fp16 bits under CLV_HALF_F16)
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;     // 16x16 MFMA accumulator
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;   // 32x32 MFMA accumulator

struct __attribute__((aligned(16))) u16x8 { uint16_t v[8]; };
struct __attribute__((aligned(8))) u16x4 { uint16_t v[4]; };
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));

#ifdef CLV_HALF_F16
#define CLV_HALF_IS_F16 1
#define CLV_ONE_PAIR 0x3c003c00u          // (1.0, 1.0) as a packed pair
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 bf16x8_t;  // MFMA A/B operand (4 VGPRs)
typedef _Float16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf2f(bf16_t h) {
    union { bf16_t s; _Float16 f; } c;
    c.s = h;
    return (float)c.f;
}
// the low / high 16-bit element of a packed pair as fp32 (bf16: a shift / a mask; f16: v_cvt_f32_f16 with op_sel)
__device__ __forceinline__ float half_lo(uint32_t w) { return bf2f((bf16_t)(w & 0xffffu)); }
__device__ __forceinline__ float half_hi(uint32_t w) { return bf2f((bf16_t)(w >> 16)); }
#else
#define CLV_HALF_IS_F16 0
#define CLV_ONE_PAIR 0x3f803f80u
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;  // MFMA A/B operand (4 VGPRs)
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ float half_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float half_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif

// fp32 -> 16-bit element, round-to-nearest-even, in hardware: one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 per PAIR (gfx950).
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const f32x2_hw_t v = {lo, hi};
    union { bf16x2_hw_t b; uint32_t u; } c;
    c.b = __builtin_convertvector(v, bf16x2_hw_t);
    return c.u;
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

union Frag8 {  // 8 elements = one MFMA 16x16x32 A/B operand
    bf16x8_t v;
    uint4 u4;
    uint2 u2[2];
    uint32_t u[4];
    uint16_t h[8];
};

// D = A(16x32) * B(32x16) + C.  Lane l supplies A[row l&15][k (l>>4)*8..+8] and
// B[k (l>>4)*8..+8][col l&15]; receives D[row (l>>4)*4+r][col l&15], r = 0..3.
__device__ __forceinline__ f32x4_t mfma16(const Frag8& a, const Frag8& b, f32x4_t c) {
#ifdef CLV_HALF_F16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a.v, b.v, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, c, 0, 0, 0);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// reduce over the 4 lane groups that share l&15 (lanes l, l^16, l^32, l^48)
__device__ __forceinline__ float grp4_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float grp4_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
// reduce over the 16 lanes that share l>>4
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

// Counter-based dropout mask: a pure function of (seed, (group, head, query) row id, key), so the
// backward kernels regenerate exactly the forward's mask.  Returns 1/(1-p) (kept) or 0 (dropped).
__device__ __forceinline__ float keep_scale(unsigned long long seed, unsigned rowid, unsigned key, unsigned thresh,
                                            float inv_keep) {
    unsigned x = (rowid * 0x9E3779B1u) ^ (key * 0x85EBCA77u) ^ (unsigned)seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    x += (unsigned)(seed >> 32);
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
    return (x >= thresh) ? inv_keep : 0.f;
}

// erf-GELU with erf by Abramowitz-Stegun 7.1.26 (|abs err| < 1.5e-7, far below bf16 resolution):
// 1 rcp + 1 exp + ~10 fma instead of libm's erff — matters where GELU sits in a GEMM epilogue.
// Phi(x) = 0.5 (1 + erf(x / sqrt2)); with z = |x| / sqrt2:  erf(z) = 1 - poly(t) exp(-z^2), t = 1 / (1 + p z),
// and exp(-z^2) = exp(-x^2 / 2) is exactly the factor the derivative needs as well.
// 1 / d as ONE v_rcp_f32 (1 ulp).  __frcp_rn expands to the IEEE division sequence (v_div_scale x 2, v_rcp, v_div_fmas,
// v_div_fixup and five fma: ~10 VALU instructions) — in a GELU epilogue that was a quarter of the instructions per element.
__device__ __forceinline__ float fast_rcp(float d) { return __builtin_amdgcn_rcpf(d); }

__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& ex) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = fast_rcp(fmaf(0.3275911f, z, 1.0f));
    ex = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);   // exp(-x^2/2) as ONE v_exp_f32 (no range fix-ups)
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float h = 0.5f * poly * t * ex;              // 0.5 (1 - erf(z))
    cdf = (x >= 0.f) ? 1.0f - h : h;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, ex;
    gelu_parts(x, cdf, ex);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, ex;
    gelu_parts(x, cdf, ex);
    return fmaf(x * 0.3989422804014327f, ex, cdf);
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// For bf16 OUTPUTS (the one-kernel MLP, fused_mlp.hip: GELU is its VALU bound — 77 M evaluations per launch):
// h(u) = 1 - Phi(u) = 0.5 erfc(u / sqrt2), u = min(|x|, 5.5), as exp2 of a degree-6 polynomial fit of log2 h on [0, 5.5]
// (Chebyshev least squares; evaluated in fp32: |h - exact| <= 2.6e-5, relative 6.5e-5, |u (h - exact)| <= 3.3e-6 — three
// orders below the bf16 rounding of the result; beyond 5.5 h stays at 1.9e-8).  ONE transcendental and 7 fma where the
// Abramowitz-Stegun form above takes v_rcp + v_exp + 10: 13 VALU slots per element instead of 20.
__device__ __forceinline__ float gelu_h(float x) {
    const float u = __builtin_amdgcn_fmed3f(fabsf(x), 0.f, 5.5f);     // (one v_med3_f32: fminf canonicalises first — two)
    float p = fmaf(2.641310222e-05f, u, -6.636010571e-04f);
    p = fmaf(p, u, 7.492294232e-03f);
    p = fmaf(p, u, -5.193681061e-02f);
    p = fmaf(p, u, -4.604588278e-01f);
    p = fmaf(p, u, -1.150443222e+00f);
    p = fmaf(p, u, -1.000073591e+00f);
    return __builtin_amdgcn_exp2f(p);
}
// two elements at a time: the Horner chain on v_pk_fma_f32 (6 instructions per PAIR)
__device__ __forceinline__ f32x2_t gelu_h2(f32x2_t x) {
    const f32x2_t u = {__builtin_amdgcn_fmed3f(fabsf(x.x), 0.f, 5.5f), __builtin_amdgcn_fmed3f(fabsf(x.y), 0.f, 5.5f)};
#define CLV_PK(c) ((f32x2_t){c, c})
    f32x2_t p = __builtin_elementwise_fma(CLV_PK(2.641310222e-05f), u, CLV_PK(-6.636010571e-04f));
    p = __builtin_elementwise_fma(p, u, CLV_PK(7.492294232e-03f));
    p = __builtin_elementwise_fma(p, u, CLV_PK(-5.193681061e-02f));
    p = __builtin_elementwise_fma(p, u, CLV_PK(-4.604588278e-01f));
    p = __builtin_elementwise_fma(p, u, CLV_PK(-1.150443222e+00f));
    p = __builtin_elementwise_fma(p, u, CLV_PK(-1.000073591e+00f));
#undef CLV_PK
    return (f32x2_t){__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
}
// GELU(x) = x Phi(x) = max(x, 0) - |x| h   (x >= 0: x (1 - h); x < 0: x h): no compare / select
__device__ __forceinline__ float gelu_fast(float x) {
    return fmaf(-fabsf(x), gelu_h(x), __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f));
}
__device__ __forceinline__ f32x2_t gelu_fast2(f32x2_t x) {
    const f32x2_t h = gelu_h2(x);
    // |x| clamped at 5.5 in place of |x|: beyond it h = 1.9e-8 — and the compiler gets a plain packed fma with a negated
    // operand (an unclamped -|x| needs a v_or per element to set the sign bits: VOP3P has no abs modifier)
    const f32x2_t u = {__builtin_amdgcn_fmed3f(fabsf(x.x), 0.f, 5.5f), __builtin_amdgcn_fmed3f(fabsf(x.y), 0.f, 5.5f)};
    const f32x2_t r = {__builtin_amdgcn_fmed3f(x.x, 0.f, 3.0e38f), __builtin_amdgcn_fmed3f(x.y, 0.f, 3.0e38f)};
    return __builtin_elementwise_fma(-u, h, r);
}
// (GELU(x), GELU'(x)):  Phi(x) = 0.5 + copysign(0.5 - h, x),  GELU' = Phi + x phi,  phi = exp(-x^2 / 2) / sqrt(2 pi)
__device__ __forceinline__ void gelu_fast_both(float x, float& act, float& grad) {
    const float h = gelu_h(x);
    const float e = __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);
    const float cdf = 0.5f + copysignf(0.5f - h, x);
    act = x * cdf;
    grad = fmaf(x * 0.3989422804014327f, e, cdf);
}
__device__ __forceinline__ void gelu_fast_both2(f32x2_t x, f32x2_t& act, f32x2_t& grad) {
    const f32x2_t h = gelu_h2(x);
    const f32x2_t xx = x * x * (f32x2_t){-0.72134752044448170f, -0.72134752044448170f};
    const f32x2_t e = {__builtin_amdgcn_exp2f(xx.x), __builtin_amdgcn_exp2f(xx.y)};
    const f32x2_t cdf = {0.5f + copysignf(0.5f - h.x, x.x), 0.5f + copysignf(0.5f - h.y, x.y)};
    act = x * cdf;
    grad = __builtin_elementwise_fma(x * (f32x2_t){0.3989422804014327f, 0.3989422804014327f}, e, cdf);
}

// Two-at-a-time versions on packed fp32 math (v_pk_mul_f32 / v_pk_fma_f32: two lanes' worth of polynomial per
// instruction; rcp / exp2 / selects stay scalar) for the GELU epilogues of the GEMM kernels, whose VALU work otherwise
// equals their MFMA time at the stage-2 widths.
__device__ __forceinline__ void gelu_parts2(f32x2_t x, f32x2_t& cdf, f32x2_t& ex) {
    const f32x2_t ax = {fabsf(x.x), fabsf(x.y)};
    const f32x2_t d = __builtin_elementwise_fma(ax, (f32x2_t){0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f},
                                                (f32x2_t){1.0f, 1.0f});
    const f32x2_t t = {fast_rcp(d.x), fast_rcp(d.y)};
    const f32x2_t xx = x * x * (f32x2_t){-0.72134752044448170f, -0.72134752044448170f};
    ex = (f32x2_t){__builtin_amdgcn_exp2f(xx.x), __builtin_amdgcn_exp2f(xx.y)};
    f32x2_t poly = __builtin_elementwise_fma((f32x2_t){1.061405429f, 1.061405429f}, t, (f32x2_t){-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2_t){1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2_t){-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(poly, t, (f32x2_t){0.254829592f, 0.254829592f});
    const f32x2_t h = poly * t * ex * (f32x2_t){0.5f, 0.5f};             // 0.5 (1 - erf(z))
    const f32x2_t oh = (f32x2_t){1.0f, 1.0f} - h;
    cdf = (f32x2_t){x.x >= 0.f ? oh.x : h.x, x.y >= 0.f ? oh.y : h.y};
}
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
    f32x2_t cdf, ex;
    gelu_parts2(x, cdf, ex);
    return x * cdf;
}
__device__ __forceinline__ f32x2_t gelu_erf_grad2(f32x2_t x) {
    f32x2_t cdf, ex;
    gelu_parts2(x, cdf, ex);
    return __builtin_elementwise_fma(x * (f32x2_t){0.3989422804014327f, 0.3989422804014327f}, ex, cdf);
}

static inline int clv_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CLV_OK : CLV_ERR_LAUNCH;
}
