// Row-wise OCP e4m3 quantisation for the fp8 GEMM path (clv_gemm_nt_fp8; BASELINE config 5: "fp8 MFMA QKV / patch-proj
// path").  q[r][k] = e4m3(x[r][k] / s[r]),  s[r] = max_k |x[r][k]| / 448  (448 = largest finite e4m3; an all-zero row gets
// s = 1).  One wave per row, the row held in registers between the max and the conversion (K <= 4096), 16-byte loads,
// 8-byte stores.  Serves both operands: activations [tokens][K] and weights [N][K] (one scale per output channel).
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int Q_MAXCH = 8;              // 16-byte chunks per lane: K <= 64 * 8 * 8 = 4096

__global__ void __launch_bounds__(256) quant_fp8_rows_kernel(const bf16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                             float* __restrict__ scale, int64_t rows, int K, int64_t ldx,
                                                             int64_t ldq) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16_t* xp = x + r * ldx;
    uint4 v[Q_MAXCH];
    float amax = 0.f;
#pragma unroll
    for (int cidx = 0; cidx < Q_MAXCH; ++cidx) {
        const int k = (cidx * 64 + lane) * 8;
        v[cidx] = make_uint4(0, 0, 0, 0);
        if (k < K) {
            v[cidx] = *reinterpret_cast<const uint4*>(xp + k);
            const uint32_t w[4] = {v[cidx].x, v[cidx].y, v[cidx].z, v[cidx].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                amax = fmaxf(amax, fabsf(half_lo(w[e])));
                amax = fmaxf(amax, fabsf(half_hi(w[e])));
            }
        }
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = amax > 0.f ? 448.0f / amax : 1.0f;
    if (lane == 0) scale[r] = s;
    uint8_t* qp = q + r * ldq;
#pragma unroll
    for (int cidx = 0; cidx < Q_MAXCH; ++cidx) {
        const int k = (cidx * 64 + lane) * 8;
        if (k < K) {
            const uint32_t w[4] = {v[cidx].x, v[cidx].y, v[cidx].z, v[cidx].w};
            float f[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f[2 * e] = fminf(fmaxf(half_lo(w[e]) * inv, -448.f), 448.f);
                f[2 * e + 1] = fminf(fmaxf(half_hi(w[e]) * inv, -448.f), 448.f);
            }
            int lo = 0, hi = 0;
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
            lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
            hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
            *reinterpret_cast<uint2*>(qp + k) = make_uint2((unsigned)lo, (unsigned)hi);
        }
    }
}

}  // namespace

extern "C" int clv_quant_fp8_rows(const void* x, void* q, float* scale, int64_t rows, int32_t K, int64_t ldx, int64_t ldq,
                                  void* stream) {
    if (!x || !q || !scale || rows < 0 || K <= 0) return CLV_ERR_ARG;
    if (rows == 0) return CLV_OK;
    if (K % 8 || K > 64 * 8 * Q_MAXCH) return CLV_ERR_UNSUPPORTED;
    if ((ldx & 7) || (ldq & 7) || ldx < K || ldq < K || (((uintptr_t)x) & 15) || (((uintptr_t)q) & 7)) return CLV_ERR_ARG;
    hipLaunchKernelGGL(quant_fp8_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (uint8_t*)q, scale, rows, (int)K, ldx, ldq);
    return clv_check_launch();
}
