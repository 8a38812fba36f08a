// Weight / bias gradient of a Linear layer for the token-parallel (huge-M, small N x K) shapes of
// the Swin stages and the patch embedding:
//       dW[n][k] += sum_m dY[m][n] * X[m][k]        db[n] += sum_m dY[m][n]
// A library GEMM tiles the N x K OUTPUT (e.g. 288 x 96 -> ~10 workgroups on 256 CUs) and walks
// M = 200 704 serially.  Here the contraction dimension is split over the whole chip: every
// workgroup owns a 128 x 128 output tile for one M-slice, streams dY and X once (16-B coalesced
// rows into LDS), feeds both MFMA operands with LDS transpose reads (the contraction index m runs
// along the ROWS of both tiles), and writes an fp32 partial; a second streaming kernel folds the
// partials into dW/db.  HBM-bound: algorithmic bytes = M (N + K) * 2.
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

constexpr int WG_THREADS = 256;
constexpr int TN = 128, TK = 128, TM = 64, LD = 128 + 8;
constexpr int CHUNKS = TM * (128 / 8) / WG_THREADS;       // 16-B chunks per thread per tile (= 4)

typedef short v4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr4(const bf16_t* base, int row0, int c0, int lr) {
    const bf16_t* p = base + (row0 + (lr >> 2)) * LD + c0 + (lr & 3) * 4;
    const v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
    union { v4s_t v; uint2 u; } cv;
    cv.v = r;
    return cv.u;
}

// global -> registers: this thread's CHUNKS 16-B pieces of a 64 x 128 tile (rows >= m_end, cols >= ncols: 0)
__device__ __forceinline__ void fetch_tile(uint4 (&v)[CHUNKS], const bf16_t* __restrict__ src, int64_t m0,
                                           int64_t m_end, int c0, int ncols, int ld, int tid) {
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int idx = tid + i * WG_THREADS;
        const int r = idx >> 4, c8 = idx & 15;
        const int64_t m = m0 + r;
        const int c = c0 + c8 * 8;
        v[i] = (m < m_end && c < ncols) ? *reinterpret_cast<const uint4*>(src + m * ld + c) : make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void store_tile(bf16_t* dst, const uint4 (&v)[CHUNKS], int tid) {
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int idx = tid + i * WG_THREADS;
        *reinterpret_cast<uint4*>(dst + (idx >> 4) * LD + (idx & 15) * 8) = v[i];
    }
}
// same, but rows are standardised on the way: x_hat = (x - mean[m]) * rstd[m]  (the LayerNorm output that
// the fused LN+GEMM forward never wrote to HBM)
__device__ __forceinline__ void store_tile_std(bf16_t* dst, const uint4 (&v)[CHUNKS], int tid, const float* mean,
                                               const float* rstd, int64_t m0, int64_t m_end) {
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int idx = tid + i * WG_THREADS;
        const int64_t m = m0 + (idx >> 4);
        Frag8 f;
        f.u4 = v[i];
        if (m < m_end) {
            const float mu = mean[m], rs = rstd[m];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                f.u[e] = pack2bf((bf2f(f.h[2 * e]) - mu) * rs, (bf2f(f.h[2 * e + 1]) - mu) * rs);
        }
        *reinterpret_cast<uint4*>(dst + (idx >> 4) * LD + (idx & 15) * 8) = f.u4;
    }
}

// Register-staged software pipeline: the global loads of tile t+1 are in flight while tile t is
// consumed from LDS (T14 "issue early / write late"); one LDS buffer, two barriers per 64 rows.
__global__ void __launch_bounds__(WG_THREADS) wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           float* __restrict__ partial, int64_t M, int N, int K,
                                                           int ldy, int ldx, int tilesK, int64_t rows_per_split,
                                                           int want_bias, const float* __restrict__ xmean,
                                                           const float* __restrict__ xrstd) {
    __shared__ __attribute__((aligned(16))) bf16_t dYs[TM * LD];
    __shared__ __attribute__((aligned(16))) bf16_t Xs[TM * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    const int tile = blockIdx.x, split = blockIdx.y;
    const int tn = tile / tilesK, tk = tile - tn * tilesK;
    const int n0 = tn * TN, k0 = tk * TK;
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;            // this wave's 64 x 64 sub-tile
    const int64_t m_begin = (int64_t)split * rows_per_split;
    int64_t m_end = m_begin + rows_per_split;
    if (m_end > M) m_end = M;
    const bool do_bias = want_bias && tk == 0 && wk == 0;

    f32x4_t acc[4][4];
    f32x4_t bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    Frag8 ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = 0x3f803f80u;      // bf16 1.0 pairs

    uint4 ry[CHUNKS], rx[CHUNKS];
    fetch_tile(ry, dy, m_begin, m_end, n0, N, ldy, tid);
    fetch_tile(rx, x, m_begin, m_end, k0, K, ldx, tid);
    for (int64_t m0 = m_begin; m0 < m_end; m0 += TM) {
        __syncthreads();                                   // previous tile fully consumed
        store_tile(dYs, ry, tid);
        if (xmean) store_tile_std(Xs, rx, tid, xmean, xrstd, m0, m_end);
        else store_tile(Xs, rx, tid);
        __syncthreads();
        if (m0 + TM < m_end) {                             // next tile's loads fly under the MFMAs below
            fetch_tile(ry, dy, m0 + TM, m_end, n0, N, ldy, tid);
            fetch_tile(rx, x, m0 + TM, m_end, k0, K, ldx, tid);
        }
#pragma unroll
        for (int h = 0; h < TM / 32; ++h) {
            Frag8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i].u2[0] = tr4(dYs, h * 32 + lg * 4, wn + i * 16, lr);        // A[n][kappa] = dY[m(kappa)][n]
                a[i].u2[1] = tr4(dYs, h * 32 + 16 + lg * 4, wn + i * 16, lr);
                b[i].u2[0] = tr4(Xs, h * 32 + lg * 4, wk + i * 16, lr);         // B[kappa][k] = X[m(kappa)][k]
                b[i].u2[1] = tr4(Xs, h * 32 + 16 + lg * 4, wk + i * 16, lr);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
                if (do_bias) bacc[i] = mfma16(a[i], ones, bacc[i]);
            }
        }
    }
    // partial[split] = [N*K dW | N db];  acc[i][j][r] = dW[n0 + wn + i*16 + lg*4 + r][k0 + wk + j*16 + lr]
    const int64_t E2 = (int64_t)N * K + N;
    float* pw = partial + (int64_t)split * E2;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn + i * 16 + lg * 4 + r;
            if (n >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + wk + j * 16 + lr;
                if (k < K) pw[(int64_t)n * K + k] = acc[i][j][r];
            }
            if (do_bias && lr == 0) pw[(int64_t)N * K + n] = bacc[i][r];
        }
}

// dw[e] += sum_s partial[s][e] (e < NK), db[e - NK] += ... (NK <= e < NK + N).  Block = 64 consecutive e x
// 16 split-lanes (1024 threads): coalesced 256-B reads, 16-way parallel splits loop, LDS tree at the end.
__global__ void __launch_bounds__(1024) fold_partials_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                             float* __restrict__ db, int64_t NK, int64_t E2,
                                                             int splits) {
    __shared__ float sh[16][64];
    const int el = threadIdx.x & 63, sp = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + el;
    const int64_t Eeff = db ? E2 : NK;
    float a = 0.f;
    if (e < Eeff)
        for (int s = sp; s < splits; s += 16) a += partial[(int64_t)s * E2 + e];
    sh[sp][el] = a;
    __syncthreads();
    if (sp == 0 && e < Eeff) {
#pragma unroll
        for (int k = 1; k < 16; ++k) a += sh[k][el];
        if (e < NK) dw[e] += a;
        else db[e - NK] += a;
    }
}

// db[n] += sum_m dy[m][n] for the library-GEMM layers (M of a few hundred..thousand rows): block = 64
// columns (8 lanes x 16 B) x 32 row-lanes, rows additionally split over blockIdx.y; LDS tree + atomics.
__global__ void __launch_bounds__(256) colsum_kernel(const bf16_t* __restrict__ dy, float* __restrict__ db, int64_t M,
                                                     int N, int ld) {
    __shared__ float sh[32][65];
    const int c8 = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int col = blockIdx.x * 64 + c8 * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (col < N) {
        for (int64_t m = (int64_t)blockIdx.y * 32 + rl; m < M; m += (int64_t)gridDim.y * 32) {
            Frag8 v;
            v.u4 = *reinterpret_cast<const uint4*>(dy + m * ld + col);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v.h[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[rl][c8 * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) a += sh[r][threadIdx.x];
        const int n = blockIdx.x * 64 + threadIdx.x;
        if (n < N) atomicAdd(db + n, a);
    }
}

int pick_splits(int64_t M, int tiles) {
    int64_t s = (768 + tiles - 1) / tiles;             // ~3 workgroups per CU
    const int64_t max_by_rows = (M + 255) / 256;        // >= 256 rows per slice
    if (s > max_by_rows) s = max_by_rows;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

}  // namespace

extern "C" int64_t clv_linear_wgrad_work_floats(int64_t M, int32_t N, int32_t K) {
    const int tiles = ((N + TN - 1) / TN) * ((K + TK - 1) / TK);
    const int splits = pick_splits(M, tiles);
    return (int64_t)splits * ((int64_t)N * K + N);
}

extern "C" int clv_linear_wgrad(const void* dy, const void* x, float* dw, float* db, float* work, int64_t M,
                                int32_t N, int32_t K, int32_t ldy, int32_t ldx, const float* xmean,
                                const float* xrstd, void* stream) {
    if ((xmean == nullptr) != (xrstd == nullptr)) return CLV_ERR_ARG;
    if (!dy || !x || !dw || !work || M <= 0 || N <= 0 || K <= 0 || (N & 7) || (K & 7) || (ldy & 7) || (ldx & 7))
        return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int tilesN = (N + TN - 1) / TN, tilesK = (K + TK - 1) / TK;
    const int tiles = tilesN * tilesK;
    const int splits = pick_splits(M, tiles);
    int64_t rows = (M + splits - 1) / splits;
    rows = (rows + TM - 1) / TM * TM;
    hipLaunchKernelGGL(wgrad_kernel, dim3(tiles, splits), dim3(WG_THREADS), 0, st, (const bf16_t*)dy,
                       (const bf16_t*)x, work, M, (int)N, (int)K, (int)ldy, (int)ldx, tilesK, rows, db ? 1 : 0, xmean, xrstd);
    int rc = clv_check_launch();
    if (rc) return rc;
    const int64_t NK = (int64_t)N * K, E2 = NK + N;
    const int64_t Eeff = db ? E2 : NK;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)((Eeff + 63) / 64)), dim3(1024), 0, st, work, dw, db, NK, E2,
                       splits);
    return clv_check_launch();
}

extern "C" int clv_colsum(const void* dy, float* db, int64_t M, int32_t N, int32_t ld, void* stream) {
    if (!dy || !db || M <= 0 || N <= 0 || (N & 7) || (ld & 7)) return CLV_ERR_ARG;
    const int xb = (N + 63) / 64;
    int64_t ys = (M + 63) / 64;                      // >= 2 rows per row-lane per block
    const int64_t want = (512 + xb - 1) / xb;
    if (ys > want) ys = want;
    if (ys < 1) ys = 1;
    hipLaunchKernelGGL(colsum_kernel, dim3(xb, (unsigned)ys), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, db,
                       M, (int)N, (int)ld);
    return clv_check_launch();
}
