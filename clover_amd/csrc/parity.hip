// fp32 GEMM of the parity mode (gfx950): C = A . B^T + bias on the exact-f32 matrix instruction
// (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, MI355X_MICROARCH.md "Matrix cores").
//
// The parity mode evaluates the Clover step with fp32 storage and fp32 arithmetic through the same host graph and
// the same index logic as the bf16 path, so that its losses can be compared with the reference's CPU fp32 path at the
// north-star tolerance (1e-3) — see clover_amd/parity.py and tests/test_parity_gpu.py.  This kernel stands in for every
// Linear of the path there (torch.nn.Linear at swin_transformer_3d.py:257-259,361-366,527, transformers' BertSelfAttention /
// BertIntermediate / BertOutput, mlm_itm_head.py:33-41).  It is a plain LDS-tiled kernel (64 x 64 x 16 tiles, 4 waves,
// 32 x 32 per wave): throughput is not its job.
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int SG_BM = 64, SG_BN = 64, SG_BK = 16, SG_LD = SG_BK + 1;

template <bool VEC>
__global__ void __launch_bounds__(256) sgemm_tiled_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          const float* __restrict__ bias, float* __restrict__ C, int M,
                                                          int N, int K, int64_t lda, int64_t ldb, int64_t ldc) {
    __shared__ float As[SG_BM * SG_LD], Bs[SG_BN * SG_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lg = lane >> 4, lr = lane & 15;
    const int m0 = blockIdx.y * SG_BM, n0 = blockIdx.x * SG_BN;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int srow = tid >> 2, sk = (tid & 3) * 4;            // staging: one float4 of A and of B per thread
    for (int k0 = 0; k0 < K; k0 += SG_BK) {
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
        const int am = m0 + srow, bn = n0 + srow;
        if (VEC) {
            if (am < M) av = *reinterpret_cast<const float4*>(A + am * lda + k0 + sk);
            if (bn < N) bv = *reinterpret_cast<const float4*>(B + bn * ldb + k0 + sk);
        } else {
            float* a4 = reinterpret_cast<float*>(&av);
            float* b4 = reinterpret_cast<float*>(&bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = k0 + sk + e;
                if (am < M && kk < K) a4[e] = A[am * lda + kk];
                if (bn < N && kk < K) b4[e] = B[bn * ldb + kk];
            }
        }
        __syncthreads();                                      // the previous step's fragment reads are done
        float* as = As + srow * SG_LD + sk;
        float* bs = Bs + srow * SG_LD + sk;
        as[0] = av.x; as[1] = av.y; as[2] = av.z; as[3] = av.w;
        bs[0] = bv.x; bs[1] = bv.y; bs[2] = bv.z; bs[3] = bv.w;
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < SG_BK; k4 += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[(wr + i * 16 + lr) * SG_LD + k4 + lg];
                b[i] = Bs[(wc + i * 16 + lr) * SG_LD + k4 + lg];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // acc[i][j][r] = C[m0 + wr + i*16 + lg*4 + r][n0 + wc + j*16 + lr]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int cn = n0 + wc + j * 16 + lr;
            if (cn >= N) continue;
            const float bb = bias ? bias[cn] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cm = m0 + wr + i * 16 + lg * 4 + r;
                if (cm < M) C[cm * ldc + cn] = acc[i][j][r] + bb;
            }
        }
}

// General-stride fp32 GEMM for the [batch, D]-sized projection heads (ssl_head.py:24-35,158-166,240-245: a handful of
// rows): C[i][j] (ldc) (+)= sum_k A(i,k) B(j,k) (+ bias[j]),  A(i,k) = A[i*sai + k*sak],  B(j,k) = B[j*sbj + k*sbk] — one
// wave per 16 x 16 tile on the exact-f32 MFMA, so that forward (x W^T), input gradient (dy W) and weight gradient
// (dy^T x, accumulated into the fp32 gradient slab) of such a layer are the SAME kernel with different strides.
template <int SS_WAVES>
__global__ void __launch_bounds__(64 * SS_WAVES) sgemm_strided_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           const float* __restrict__ bias, float* __restrict__ C, int M,
                                                           int N, int K, int64_t sai, int64_t sak, int64_t sbj, int64_t sbk,
                                                           int64_t ldc, int accumulate, float* __restrict__ rowsum) {
    // 8 waves share one 16 x 16 output tile and split the contraction (these layers have 8-64 rows: a tile per wave left
    // under a hundred waves walking K = 768..1536 alone: 29 us per launch); partial tiles meet in LDS.  The weight-gradient
    // form of the same layers contracts over those 8-64 rows and has 2 304 tiles: one wave per tile (SS_WAVES = 1).
    __shared__ float red[SS_WAVES > 1 ? SS_WAVES - 1 : 1][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    const int i = blockIdx.y * 16 + lr, j = blockIdx.x * 16 + lr;
    const bool av = i < M, bv = j < N;
    const float* ap = A + (int64_t)(av ? i : 0) * sai;
    const float* bp = B + (int64_t)(bv ? j : 0) * sbj;
    const int kchunk = ((K + SS_WAVES * 16 - 1) / (SS_WAVES * 16)) * 16;     // per wave, a multiple of 16
    const int kb = wave * kchunk, ke = min(K, kb + kchunk);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float asum = 0.f;                                         // rowsum: sum_k A(i, k) (the bias gradient of the weight-gradient form)
    constexpr int KT = 8;                                     // scalar form: two 16-deep blocks per trip, 16 loads in flight
    // Both operands contiguous along the contraction (the forward form x W^T of a head layer, 16 rows against a 2.4-4.7 MB
    // fp32 weight): 16-byte loads, six per operand in flight — a lane's float4 holds k = k0 + 16 e + 4 lg .. + 3, and since A
    // and B use the SAME assignment the order of k inside the 16-deep block is immaterial to the sum of products.  The
    // scalar form below moved the weight at 130 GB/s (36 us for 1536 x 768); round 6.
    const bool vec = SS_WAVES > 1 && sak == 1 && sbk == 1 && (K & 3) == 0 && (((uintptr_t)ap | (uintptr_t)bp) & 15) == 0 &&
                     ((sai | sbj) & 3) == 0;
    if (vec) {
        constexpr int KV = 6;
        for (int k0 = kb; k0 < ke; k0 += 16 * KV) {
            float4 a4[KV], b4[KV];
#pragma unroll
            for (int e = 0; e < KV; ++e) {
                const int k = k0 + e * 16 + lg * 4;
                const bool in = k < ke;                       // ke is K or a multiple of 16; K is a multiple of 4
                a4[e] = (av && in) ? *reinterpret_cast<const float4*>(ap + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                b4[e] = (bv && in) ? *reinterpret_cast<const float4*>(bp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int e = 0; e < KV; ++e) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e].x, b4[e].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e].y, b4[e].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e].z, b4[e].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[e].w, b4[e].w, acc, 0, 0, 0);
            }
        }
    } else
    for (int k0 = kb; k0 < ke; k0 += 4 * KT) {
        float a[KT], b[KT];
#pragma unroll
        for (int e = 0; e < KT; ++e) {
            const int k = k0 + e * 4 + lg;                    // k-step e: lane group lg holds k = k0 + 4e + lg
            a[e] = (av && k < ke) ? ap[k * sak] : 0.f;
            b[e] = (bv && k < ke) ? bp[k * sbk] : 0.f;
            if (SS_WAVES == 1) asum += a[e];
        }
#pragma unroll
        for (int e = 0; e < KT; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
    if (SS_WAVES == 1 && rowsum && blockIdx.x == 0) {         // the first column tile of a row tile also owns its row sums
        asum += __shfl_xor(asum, 16);
        asum += __shfl_xor(asum, 32);
        if (lg == 0 && av) rowsum[i] = (accumulate ? rowsum[i] : 0.f) + asum;
    }
    if (SS_WAVES > 1) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][lane][r] = acc[r];
        }
        __syncthreads();
    }
    if (wave == 0 && j < N) {
#pragma unroll
        for (int w = 0; w < SS_WAVES - 1; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += red[w][lane][r];
        const float bb = bias ? bias[j] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ci = blockIdx.y * 16 + lg * 4 + r;
            if (ci < M) {
                float* cp = C + (int64_t)ci * ldc + j;
                *cp = (accumulate ? *cp : 0.f) + acc[r] + bb;
            }
        }
    }
}

}  // namespace

static int sgemm_strided_impl(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N,
                              int32_t K, int64_t sai, int64_t sak, int64_t sbj, int64_t sbk, int64_t ldc,
                              int32_t accumulate, float* rowsum, void* stream) {
    if (!A || !B || !C || M < 0 || N <= 0 || K <= 0 || ldc < N) return CLV_ERR_ARG;
    if (M == 0) return CLV_OK;
    if (M > 0x7fffffff) return CLV_ERR_UNSUPPORTED;
    if (rowsum && K > 96) return CLV_ERR_UNSUPPORTED;
    const dim3 grid((N + 15) / 16, (unsigned)((M + 15) / 16));
    if (K <= 96)                                             // a short contraction (weight-gradient form): nothing to split
        hipLaunchKernelGGL(sgemm_strided_kernel<1>, grid, dim3(64), 0, (hipStream_t)stream, A, B, bias, C, (int)M, N, K, sai,
                           sak, sbj, sbk, ldc, accumulate, rowsum);
    else
        hipLaunchKernelGGL(sgemm_strided_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, A, B, bias, C, (int)M, N, K, sai,
                           sak, sbj, sbk, ldc, accumulate, (float*)nullptr);
    return clv_check_launch();
}

extern "C" int clv_sgemm_strided(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N,
                                 int32_t K, int64_t sai, int64_t sak, int64_t sbj, int64_t sbk, int64_t ldc,
                                 int32_t accumulate, void* stream) {
    return sgemm_strided_impl(A, B, bias, C, M, N, K, sai, sak, sbj, sbk, ldc, accumulate, nullptr, stream);
}

extern "C" int clv_sgemm_strided_rowsum(const float* A, const float* B, float* C, float* rowsum, int64_t M, int32_t N,
                                        int32_t K, int64_t sai, int64_t sak, int64_t sbj, int64_t sbk, int64_t ldc,
                                        int32_t accumulate, void* stream) {
    if (!rowsum) return CLV_ERR_ARG;
    return sgemm_strided_impl(A, B, nullptr, C, M, N, K, sai, sak, sbj, sbk, ldc, accumulate, rowsum, stream);
}

extern "C" int clv_sgemm_nt(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N,
                            int32_t K, int64_t lda, int64_t ldb, int64_t ldc, void* stream) {
    if (!A || !B || !C || M < 0 || N <= 0 || K <= 0 || lda < K || ldb < K || ldc < N) return CLV_ERR_ARG;
    if (M == 0) return CLV_OK;
    if (M > 0x7fffffff) return CLV_ERR_UNSUPPORTED;
    const dim3 grid((N + SG_BN - 1) / SG_BN, (unsigned)((M + SG_BM - 1) / SG_BM));
    const bool vec = (K % SG_BK == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                     (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (reinterpret_cast<uintptr_t>(B) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(sgemm_tiled_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, A, B, bias, C, (int)M, N, K,
                           lda, ldb, ldc);
    else
        hipLaunchKernelGGL(sgemm_tiled_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, A, B, bias, C, (int)M, N, K,
                           lda, ldb, ldc);
    return clv_check_launch();
}
