// EXPERIMENTAL — not part of libclover_hip.so: measured on MI355X it equals the tuned library
// GEMMs on the step's shapes (25-40 us at 6-8 us roofline) and wins only on the narrow ones (384 x 384: 13 vs 19 us),
// so the product keeps the library path.  Kept as the starting point for a 256 x 128 / BK = 64 version.
// Build: hipcc --offload-arch=gfx950 -O3 -I.. -c gemm_tile.hip; driver: tools/probes/gemm_bench.py (needs the two
// clv_gemm_nt* entry points added back to the Makefile / header / _lib.py).
// LDS-tiled bf16 MFMA GEMMs for the mid-size Linear layers of the step (Swin stages 1-3, fusion encoder):
//        NT:  Y[M][N]  = X[M][K] * W[N][K]^T + bias          (forward of nn.Linear, optional erf-GELU)
//        NN:  dX[M][K] = dY[M][N] * W[N][K]                   (input gradient)
// with M of 3 000 .. 50 000 rows and N, K of 192 .. 3072.  These shapes are short in the contraction
// dimension (6 .. 48 steps of 32) and memory-lean (ideal time 3 .. 15 us); the library kernels the heuristics
// pick for them run at 3 - 7x their roofline.  Here: 128 x 128 output tile per workgroup (4 waves, 64 x 64 each),
// 32-deep stages of both operands brought global -> LDS by LDS-DMA into a 4-slot ring with counted vmcnt waits
// (the scheme of gemm_wgrad.hip), XOR-swizzled on the source side so that the 16-byte MFMA operand reads are
// bank-conflict free, output through an LDS tile as full 256-byte rows, the tiles of one row block pinned to one
// XCD so the block's X rows are fetched from HBM once.
#include "../common.hpp"
#include <stdint.h>

namespace {

constexpr int GT_THREADS = 256;
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int GT_RING = 4;
constexpr int OPER_BYTES = BM * BK * 2;                       // one operand stage: 128 rows x 64 B = 8 KiB
constexpr int STAGE_BYTES = 2 * OPER_BYTES;
constexpr int LDO = BN + 8;                                   // output staging row (bf16 elements)

__device__ __attribute__((aligned(16))) unsigned int g_zero16b[4];

__device__ __forceinline__ void gt_dma16(const bf16_t* src, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_byte)
                 : "memory");
}
template <int N_>
__device__ __forceinline__ void gt_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

typedef short gt_v4s __attribute__((ext_vector_type(4)));

// One stage of an operand with K-contiguous rows: LDS [128 rows][4 chunks of 16 B]; the chunk c of row r sits at
// physical chunk c ^ ((r >> 2) & 3): the 16 rows x 16 B an MFMA operand read touches then cover all 64 banks once.
// Rows >= nrows read a zero chunk.
struct RowOp {
    const bf16_t* src[2];     // this lane's two pieces (rows piece*16 + lane/4), pre-swizzled chunk
    bool ok[2];
    int64_t step;             // elements per stage (BK)
};

__device__ __forceinline__ void rowop_init(RowOp& o, const bf16_t* base, int64_t row0, int64_t nrows, int ld, int wave,
                                           int lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = wave * 2 + j;
        const int r = piece * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        o.ok[j] = row0 + r < nrows;
        o.src[j] = base + (row0 + r) * ld + c * 8;
    }
    o.step = BK;
}

__device__ __forceinline__ void rowop_issue(RowOp& o, unsigned lds_oper_base, int wave) {
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16b);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        gt_dma16(o.ok[j] ? o.src[j] : zero,
                 __builtin_amdgcn_readfirstlane(lds_oper_base + (unsigned)((wave * 2 + j) * 1024)));
        o.src[j] += o.step;
    }
}

// MFMA operand (8 consecutive k of row `r`, k chunk lg) of a K-contiguous stage
__device__ __forceinline__ Frag8 rowop_frag(const unsigned char* oper, int r, int lg) {
    Frag8 f;
    f.u4 = *reinterpret_cast<const uint4*>(oper + r * 64 + ((lg ^ ((r >> 2) & 3)) << 4));
    return f;
}

enum { GT_EPI_BIAS = 0, GT_EPI_GELU = 1 };

template <int EPI>
__global__ void __launch_bounds__(GT_THREADS, 2) gemm_nt_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                                bf16_t* __restrict__ pre_out, int64_t M, int N, int K,
                                                                int ldx, int ldy, int tilesN, int nmblk) {
    // the ring doubles as the output tile(s) after the K loop: one [128][LDO] bf16 tile, two with the GELU epilogue
    constexpr int SMEM = (EPI == GT_EPI_GELU && 2 * BM * LDO * 2 > GT_RING * STAGE_BYTES) ? 2 * BM * LDO * 2 : GT_RING * STAGE_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    // the N tiles of one row block on ONE XCD (workgroup i runs on XCD i % 8), consecutive slots
    const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
    const int mblk = xcd + 8 * (xslot / tilesN), tn = xslot % tilesN;
    if (mblk >= nmblk) return;
    const int64_t m0 = (int64_t)mblk * BM;
    const int n0 = tn * BN;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

    RowOp A, B;
    rowop_init(A, x, m0, M, ldx, wave, lane);
    rowop_init(B, w, n0, N, K, wave, lane);
    const unsigned ring_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&ring[0]);

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nst = K / BK;
    // every stage is issued as exactly 4 pieces per wave, also past the end (all-zero source), so that the counted
    // waits stay uniform
#pragma unroll
    for (int d = 0; d < GT_RING - 1; ++d) {
        if (d >= nst) { A.ok[0] = A.ok[1] = B.ok[0] = B.ok[1] = false; }
        rowop_issue(A, ring_base + d * STAGE_BYTES, wave);
        rowop_issue(B, ring_base + d * STAGE_BYTES + OPER_BYTES, wave);
    }
    int slot = 0;
    for (int st = 0; st < nst; ++st) {
        gt_wait_vm<(GT_RING - 2) * 4>();
        __builtin_amdgcn_s_barrier();
        {
            const int ns = slot == 0 ? GT_RING - 1 : slot - 1;
            if (st + GT_RING - 1 >= nst) { A.ok[0] = A.ok[1] = B.ok[0] = B.ok[1] = false; }
            rowop_issue(A, ring_base + ns * STAGE_BYTES, wave);
            rowop_issue(B, ring_base + ns * STAGE_BYTES + OPER_BYTES, wave);
        }
        const unsigned char* As = ring + slot * STAGE_BYTES;
        const unsigned char* Bs = As + OPER_BYTES;
        slot = slot == GT_RING - 1 ? 0 : slot + 1;
        Frag8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = rowop_frag(As, wm + i * 16 + lr, lg);
            b[i] = rowop_frag(Bs, wn + i * 16 + lr, lg);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);   // swapped: D[n][m], see epilogue
    }
    gt_wait_vm<0>();
    __syncthreads();                                          // ring free: reuse it as the output tile(s)

    // The MFMA ran with swapped operands, so acc[i][j][r] = Y[m0 + wm + i*16 + lr][n0 + wn + j*16 + lg*4 + r]: a lane
    // holds 4 CONSECUTIVE columns of one row -> one 8-byte LDS write per (i, j) instead of four 2-byte ones.
    bf16_t* out_s = reinterpret_cast<bf16_t*>(ring);          // [128][LDO]
    bf16_t* pre_s = out_s + BM * LDO;                         // GELU: the pre-activation tile
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + j * 16 + lg * 4;
        float bn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bn[r] = (bias && n + r < N) ? bias[n + r] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = (wm + i * 16 + lr) * LDO + wn + j * 16 + lg * 4;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bn[r];
            if (EPI == GT_EPI_GELU) {
                *reinterpret_cast<uint2*>(pre_s + off) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
            }
            *reinterpret_cast<uint2*>(out_s + off) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
        }
    }
    __syncthreads();
    for (int idx = tid; idx < BM * (BN / 8); idx += GT_THREADS) {
        const int r = idx >> 4, c8 = idx & 15;
        if (m0 + r < M && n0 + c8 * 8 < N) {
            *reinterpret_cast<uint4*>(y + (m0 + r) * ldy + n0 + c8 * 8) =
                *reinterpret_cast<const uint4*>(out_s + r * LDO + c8 * 8);
            if (EPI == GT_EPI_GELU)
                *reinterpret_cast<uint4*>(pre_out + (m0 + r) * ldy + n0 + c8 * 8) =
                    *reinterpret_cast<const uint4*>(pre_s + r * LDO + c8 * 8);
        }
    }
}

}  // namespace

extern "C" int clv_gemm_nt_supported(int64_t M, int32_t N, int32_t K) {
    return M >= 1024 && N >= 64 && N % 8 == 0 && K >= 64 && K % 32 == 0;
}

extern "C" int clv_gemm_nt(const void* x, const void* w, const float* bias, void* y, void* pre_out, int64_t M, int32_t N,
                           int32_t K, int32_t ldx, int32_t ldy, int32_t epilogue, void* stream) {
    if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || (K % BK) || (N & 7) || (ldx & 7) || (ldy & 7) || ldx < K || ldy < N)
        return CLV_ERR_ARG;
    if (epilogue == GT_EPI_GELU && !pre_out) return CLV_ERR_ARG;
    if (epilogue != GT_EPI_BIAS && epilogue != GT_EPI_GELU) return CLV_ERR_UNSUPPORTED;
    static_assert(BM * LDO * 2 <= GT_RING * STAGE_BYTES, "the output tile must fit the ring");
    hipStream_t st = (hipStream_t)stream;
    const int tilesN = (N + BN - 1) / BN;
    const int nmblk = (int)((M + BM - 1) / BM);
    const unsigned grid = (unsigned)(8 * tilesN * ((nmblk + 7) / 8));
    if (epilogue == GT_EPI_GELU)
        hipLaunchKernelGGL(gemm_nt_kernel<GT_EPI_GELU>, dim3(grid), dim3(GT_THREADS), 0, st, (const bf16_t*)x,
                           (const bf16_t*)w, bias, (bf16_t*)y, (bf16_t*)pre_out, M, (int)N, (int)K, (int)ldx, (int)ldy,
                           tilesN, nmblk);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<GT_EPI_BIAS>, dim3(grid), dim3(GT_THREADS), 0, st, (const bf16_t*)x,
                           (const bf16_t*)w, bias, (bf16_t*)y, (bf16_t*)pre_out, M, (int)N, (int)K, (int)ldx, (int)ldy,
                           tilesN, nmblk);
    return clv_check_launch();
}
