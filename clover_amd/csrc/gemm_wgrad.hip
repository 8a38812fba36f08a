// Weight / bias gradient of a Linear layer:
//       dW[n][k] += sum_m dY[m][n] * X[m][k]        db[n] += sum_m dY[m][n]
// For the token-parallel layers (Swin stages, patch embedding: M = 12 544 .. 200 704 rows, N x K small) a library
// GEMM tiles the N x K OUTPUT (e.g. 288 x 96 -> ~10 workgroups on 256 CUs) and walks M serially.  Here the
// contraction dimension is split over the whole chip: every workgroup owns a 128 x 128 output tile for one
// M-slice, streams its dY and X rows through LDS, feeds both MFMA operands with LDS transpose reads (the
// contraction index m runs along the ROWS of both tiles), and writes an fp32 partial; a second streaming
// kernel folds the partials into dW/db.  With one M-slice (M <= 1024: the text / fusion layers) the tile is
// accumulated straight into dW/db and there is no second kernel.  HBM-bound: algorithmic bytes = M (N + K) * 2.
#include "common.hpp"
#include "../../include/clover_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int WG_THREADS = 256;
constexpr int TN = 128, TK = 128, TM = 64;
constexpr int LD = 128 + 16;   // row stride 72 dwords = 8 banks: the 4 rows x 32 B of a transpose read tile the banks
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
__global__ void __launch_bounds__(WG_THREADS, 2) wgrad_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                           float* __restrict__ partial, int64_t M, int N, int K,
                                                           int ldy, int ldx, int tiles, int tilesK,
                                                           int64_t rows_per_split, int want_bias,
                                                           const float* __restrict__ xmean,
                                                           const float* __restrict__ xrstd) {
    __shared__ __attribute__((aligned(16))) bf16_t dYs[TM * LD];
    __shared__ __attribute__((aligned(16))) bf16_t Xs[TM * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
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
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = CLV_ONE_PAIR;      // 1.0 pairs

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


// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant (the default): the 32-row stages of dY and X go global -> LDS directly
// (global_load_lds_dwordx4, 1 KiB per wave-instruction) into a two-slot ring, so no staging VGPRs and
// no ds_write pass: ~110 VGPRs -> 4 workgroups per CU, each with one stage in flight while it feeds
// the MFMAs from the other.  The DMA writes LDS linearly (wave base + lane x 16 B), rows are 256 B
// (all rows on the same banks), so the 16-B chunk a lane fetches is XOR-swizzled on the SOURCE side
// and the transpose reads apply the same involution: 32-B pair p of row r lives at pair p ^ (r & 7).
constexpr int SM = 32;                                    // rows per stage
constexpr int STAGE = SM * 128;                           // bf16 elements per tensor per stage (8 KiB)
#ifndef WG_RING
#define WG_RING 4
#endif
constexpr int RING = WG_RING;                             // ring slots (16 KiB each); RING-1 stages in flight
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4];   // source of out-of-range chunks

// One LDS-DMA piece: 64 lanes x 16 B from per-lane global addresses to LDS [lds_byte .. +1 KiB), asynchronously.
// Issued from asm so that hipcc does not count it: the loop below keeps RING-1 stages in flight across its
// barriers with counted s_waitcnt vmcnt(N); the compiler's own bookkeeping would drain to vmcnt(0) at each one.
__device__ __forceinline__ void dma16(const bf16_t* src, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_byte)
                 : "memory");
}
template <int N_>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}
__device__ __forceinline__ uint2 tr4s(const bf16_t* base, int row, int c0, int lr) {
    // rows row..row+3 (lane lr>>2), columns c0 + (lr&3)*4 .. +4 of the swizzled [SM][128] stage
    const int r = row + (lr >> 2);
    const bf16_t* p = base + r * 128 + ((((c0 >> 4) ^ (r & 7)) << 4) | ((lr & 3) << 2));
    const v4s_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
    union { v4s_t v; uint2 u; } cv;
    cv.v = v;
    return cv.u;
}

template <bool ACCUM>
__global__ void __launch_bounds__(WG_THREADS, 2) wgrad_dma_kernel(const bf16_t* __restrict__ dy,
                                                                  const bf16_t* __restrict__ x,
                                                                  float* __restrict__ out, float* __restrict__ out_b,
                                                                  int64_t M, int N, int K, int ldy, int ldx, int tiles,
                                                                  int tilesK, int nsplits, int64_t rows_per_split,
                                                                  int want_bias) {
    __shared__ __attribute__((aligned(1024))) bf16_t ring[RING][2][STAGE];   // [slot][dY | X]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    // Workgroup i runs on XCD i % 8 (round-robin dispatch).  All output tiles of one M-slice are placed on ONE
    // XCD, in consecutive slots, so the dY / X rows they share come from HBM once and are re-read from that
    // XCD's L2 (the slices' workgroups start together and walk their rows at the same pace).
    // (nsplits < 0: plain order — few slices, or few tiles per slice, where spreading over all XCDs wins.)
    int split, tile;
    if (nsplits > 0) {
        const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
        split = xcd + 8 * (xslot / tiles);
        tile = xslot % tiles;
        if (split >= nsplits) return;
    } else {
        split = blockIdx.x / tiles;
        tile = blockIdx.x - split * tiles;
    }
    const int tn = tile / tilesK, tk = tile - tn * tilesK;
    const int n0 = tn * TN, k0 = tk * TK;
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    const int64_t m_begin = (int64_t)split * rows_per_split;
    int64_t m_end = m_begin + rows_per_split;
    if (m_end > M) m_end = M;
    const bool do_bias = want_bias && tk == 0 && wk == 0;

    // this lane's two chunks per tensor per stage: LDS position p = j*256 + tid -> row p>>4, physical chunk p&15
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);
    int rowj[2];
    const bf16_t *sy[2], *sx[2];
    bool vy[2], vx[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = j * 256 + tid;
        const int r = p >> 4, c = ((p & 15) ^ ((r & 7) << 1)) * 8;
        rowj[j] = r;
        vy[j] = n0 + c < N;
        vx[j] = k0 + c < K;
        sy[j] = dy + (m_begin + r) * ldy + n0 + c;
        sx[j] = x + (m_begin + r) * ldx + k0 + c;
    }
    const unsigned ring_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)&ring[0][0][0] + (unsigned)wave * 1024u);
    // every stage is issued as exactly 4 pieces per wave, also past m_end (all-zero source), so that the
    // counted waits below stay uniform
    auto issue = [&](int slot, int64_t m0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool in = m0 + rowj[j] < m_end;
            dma16((in && vy[j]) ? sy[j] : zero, ring_base + (unsigned)((slot * 2 + 0) * STAGE * 2 + j * 4096));
            dma16((in && vx[j]) ? sx[j] : zero, ring_base + (unsigned)((slot * 2 + 1) * STAGE * 2 + j * 4096));
            sy[j] += (int64_t)SM * ldy;
            sx[j] += (int64_t)SM * ldx;
        }
    };

    f32x4_t acc[4][4];
    f32x4_t bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    Frag8 ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = CLV_ONE_PAIR;

#pragma unroll
    for (int d = 0; d < RING - 1; ++d) issue(d, m_begin + d * SM);
    int slot = 0;
    for (int64_t m0 = m_begin; m0 < m_end; m0 += SM) {
        wait_vm<(RING - 2) * 4>();       // this wave's pieces of stage m0 have landed (RING-2 later stages may fly on)
        __builtin_amdgcn_s_barrier();    // ... and every other wave's; everyone is also done reading the previous slot
        issue(slot == 0 ? RING - 1 : slot - 1, m0 + (int64_t)(RING - 1) * SM);
        const bf16_t* Ys = ring[slot][0];
        const bf16_t* Xs = ring[slot][1];
        slot = slot == RING - 1 ? 0 : slot + 1;
        Frag8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i].u2[0] = tr4s(Ys, lg * 4, wn + i * 16, lr);
            a[i].u2[1] = tr4s(Ys, 16 + lg * 4, wn + i * 16, lr);
            b[i].u2[0] = tr4s(Xs, lg * 4, wk + i * 16, lr);
            b[i].u2[1] = tr4s(Xs, 16 + lg * 4, wk + i * 16, lr);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
            if (do_bias) bacc[i] = mfma16(a[i], ones, bacc[i]);
        }
    }
    wait_vm<0>();                        // drain the all-zero tail stages before the LDS is released
    // ACCUM: out = dW (+=), out_b = db (+=).  Otherwise out = this split's partial [N*K dW | N db].
    float* pw = ACCUM ? out : out + (int64_t)split * ((int64_t)N * K + N);
    float* pb = ACCUM ? out_b : pw + (int64_t)N * K;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + wn + i * 16 + lg * 4 + r;
            if (n >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + wk + j * 16 + lr;
                if (k < K) {
                    // ACCUM: this workgroup is the only writer of the element; the no-return L2 atomic is
                    // fire-and-forget, where "+=" would be 64 dependent load -> add -> store round trips
                    if (ACCUM) atomicAdd(&pw[(int64_t)n * K + k], acc[i][j][r]);
                    else pw[(int64_t)n * K + k] = acc[i][j][r];
                }
            }
            if (do_bias && lr == 0) {
                if (ACCUM) atomicAdd(&pb[n], bacc[i][r]);
                else pb[n] = bacc[i][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// Lean-loop version of wgrad_dma_kernel (same tiling, ring, swizzle and output).  The first version spent ~70 VALU and
// ~50 SALU instructions per 32-row stage on per-piece pointer selects / 64-bit increments / bounds predicates / slot
// arithmetic around its 16-20 MFMAs (rocprofv3 SQ counters: 3 445 VALU + 2 490 SALU per 49-stage wave) and, with one
// wave per SIMD, ran at ~1 700 cycles per stage.  Here:
//   * the loop is unrolled over the 4 ring slots, so every LDS address (DMA destination and transpose read) is a
//     per-lane constant plus an IMMEDIATE;
//   * the DMA uses the SGPR-base + 32-bit-VGPR-offset form: the per-lane offsets never change and a stage advance is
//     two scalar adds per tensor; the four pieces of a stage are one asm block (M0 saved / restored once);
//   * column overhang is handled by CLAMPING the source column (those products land in output rows / columns that are
//     never stored); only the row tail needs zeros, and only the last stage of a slice can be ragged: it (and the
//     drain of the pipeline) runs in a generic slow loop of at most 2 * RING stages.
__device__ __forceinline__ void dma4(unsigned lds_wave_base, unsigned voff_y0, unsigned voff_y1, unsigned voff_x0,
                                     unsigned voff_x1, const bf16_t* base_y, const bf16_t* base_x, int slot_const) {
    unsigned keep;
    // stage layout: [slot][dY | X][8 KiB], piece j at + j * 4096, this wave at + wave * 1024 (in lds_wave_base)
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_add_u32 m0, %[lds], %[o0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vy0], %[by]\n\t"
        "s_add_u32 m0, %[lds], %[o1]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vy1], %[by]\n\t"
        "s_add_u32 m0, %[lds], %[o2]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vx0], %[bx]\n\t"
        "s_add_u32 m0, %[lds], %[o3]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vx1], %[bx]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [lds] "s"(lds_wave_base), [o0] "s"(slot_const), [o1] "s"(slot_const + 4096), [o2] "s"(slot_const + 8192),
          [o3] "s"(slot_const + 8192 + 4096), [vy0] "v"(voff_y0), [vy1] "v"(voff_y1), [vx0] "v"(voff_x0),
          [vx1] "v"(voff_x1), [by] "s"(base_y), [bx] "s"(base_x)
        : "memory", "scc");
}

// Gradient-norm partial sums from the kernels that WRITE a first-touch weight gradient (overwrite bit 2; round 6): the sum of
// squares of what a workgroup stores goes, one atomic per wave, to one of CLV_SUMSQ_SLOTS accumulator slots 64 bytes apart
// (blockIdx picks the slot: ~300 atomics per slot and step instead of ~20 000 on one address); clv_optim_prep_slots adds the
// slots to the norm.  The engine's separate sumsq pass then covers only what these kernels do not write.
__device__ __forceinline__ void ssq_commit(float* ssq, float q) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    if ((threadIdx.x & 63) == 0 && q != 0.f) atomicAdd(ssq + (blockIdx.x & (CLV_SUMSQ_SLOTS - 1)) * 16, q);
}

template <bool ACCUM, bool RMW = false>
__device__ __forceinline__ void wgrad_dma2_body(bf16_t (&ring)[RING][2][STAGE], const int bid,
                                                const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x,
                                                float* __restrict__ out, float* __restrict__ out_b, int64_t M, int N, int K,
                                                int ldy, int ldx, int tiles, int tilesK, int nsplits,
                                                int64_t rows_per_split, int want_bias, int xcd_rot = 0,
                                                int overwrite = 0, float* __restrict__ ssq = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    int split, tile;
    if (nsplits > 0 && xcd_rot >= 0) {                                       // see wgrad_dma_kernel
        const int xcd = (bid - xcd_rot) & 7, xslot = bid >> 3;               // grouped launches rotate the XCDs per problem
        split = xcd + 8 * (xslot / tiles);
        tile = xslot % tiles;
        if (split >= nsplits) return;
    } else {
        split = bid / tiles;
        tile = bid - split * tiles;
        if (nsplits > 0 && split >= nsplits) return;                         // a grouped problem's padding blocks
    }
    const int tn = tile / tilesK, tk = tile - tn * tilesK;
    const int n0 = tn * TN, k0 = tk * TK;
    const int wn = (wave >> 1) * 64, wk = (wave & 1) * 64;
    const int64_t m_begin = (int64_t)split * rows_per_split;
    int64_t m_end = m_begin + rows_per_split;
    if (m_end > M) m_end = M;
    const bool do_bias = want_bias && tk == 0 && wk == 0;
    const int rows = (int)(m_end - m_begin);
    const int nst = (rows + SM - 1) / SM;                   // stages of this slice
    const int nfull = rows / SM;                            // ... of which complete (all 32 rows inside)
    // Wave-specialised launch (blockDim = 320): a FIFTH wave issues every LDS-DMA piece of the workgroup, the four MFMA
    // waves only read and multiply.  A wave is blocked ~60 cycles per piece it issues (the CU's vector-memory path takes
    // 1 KiB per ~16 cycles) and feeds no MFMA meanwhile — tools/probes/gemm_lab.cpp measured issue ~ compute ~ 40 % of a
    // stage each for this loop structure; on separate waves the two overlap.
    const bool ws = blockDim.x > WG_THREADS;
    if (ws && wave == 4) {
        const int ln = lane;
        unsigned py[8], px[8];
        int prow[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {                       // q = j * 4 + w: the piece wave w issued as its j-th
            const int pp = (q >> 2) * 256 + (q & 3) * 64 + ln;
            const int r = pp >> 4, c = ((pp & 15) ^ ((r & 7) << 1)) * 8;
            int cy = n0 + c, cx = k0 + c;
            cy = cy < N ? cy : N - 8;
            cx = cx < K ? cx : K - 8;
            prow[q] = r;
            py[q] = (unsigned)((r * ldy + cy) * 2);
            px[q] = (unsigned)((r * ldx + cx) * 2);
        }
        const bf16_t* pby = dy + m_begin * ldy;
        const bf16_t* pbx = x + m_begin * ldx;
        const int64_t sy = (int64_t)SM * ldy, sx = (int64_t)SM * ldx;
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(
            (unsigned)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)&ring[0][0][0]);
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);
        auto issue_p = [&](int s, int slot) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool in = s * SM + prow[q] < rows;
                const unsigned dst = lds0 + (unsigned)((q & 3) * 1024 + (q >> 2) * 4096);
                dma16(in ? reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pby) + py[q]) : zero,
                      dst + (unsigned)((slot * 2 + 0) * STAGE * 2));
                dma16(in ? reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pbx) + px[q]) : zero,
                      dst + (unsigned)((slot * 2 + 1) * STAGE * 2));
            }
            pby += sy;
            pbx += sx;
        };
        // complete stages go out in the lean form (SGPR bases + fixed lane offsets, four pieces per asm block), the ragged
        // last one through the predicated per-piece path
        auto issue_f = [&](int s, int slot) {
            if (s < nfull) {
                const int sc = slot * 2 * STAGE * 2;
#pragma unroll
                for (int w = 0; w < 4; ++w) dma4(lds0 + (unsigned)(w * 1024), py[w], py[4 + w], px[w], px[4 + w], pby, pbx, sc);
                pby += sy;
                pbx += sx;
            } else {
                issue_p(s, slot);
            }
        };
        for (int d = 0; d < RING - 1 && d < nst; ++d) issue_f(d, d);
        int nslot = (RING - 1) % RING;
        for (int s = 0; s < nst; ++s) {
            if (nst - 1 - s >= RING - 2) wait_vm<(RING - 2) * 16>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            if (s + RING - 1 < nst) {
                issue_f(s + RING - 1, nslot);
                nslot = nslot == RING - 1 ? 0 : nslot + 1;
            }
        }
        wait_vm<0>();
        return;
    }

    // per-lane source offsets (bytes from the stage's first row), fixed for the whole slice: LDS position
    // p = j*256 + tid -> row p>>4, physical chunk p&15 holds logical chunk (p&15) ^ ((row&7)<<1); columns clamped
    unsigned vy[2], vx[2];
    int rowj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = j * 256 + tid;
        const int r = p >> 4, c = ((p & 15) ^ ((r & 7) << 1)) * 8;
        int cy = n0 + c, cx = k0 + c;
        cy = cy < N ? cy : N - 8;
        cx = cx < K ? cx : K - 8;
        rowj[j] = r;
        vy[j] = (unsigned)((r * ldy + cy) * 2);
        vx[j] = (unsigned)((r * ldx + cx) * 2);
    }
    const bf16_t* by = dy + m_begin * ldy;                  // wave-uniform stage bases (SGPR pairs)
    const bf16_t* bx = x + m_begin * ldx;
    const int64_t step_y = (int64_t)SM * ldy, step_x = (int64_t)SM * ldx;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)&ring[0][0][0] + (unsigned)wave * 1024u);
    int issued = 0;                                         // stages issued so far
    // generic issue (ragged last stage: rows past m_end read a zero chunk; past nst: nothing)
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);
    auto issue_slow = [&](int slot) {
        if (issued < nst) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool in = issued * SM + rowj[j] < rows;
                dma16(in ? reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(by) + vy[j]) : zero,
                      lds_wave + (unsigned)((slot * 2 + 0) * STAGE * 2 + j * 4096));
                dma16(in ? reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(bx) + vx[j]) : zero,
                      lds_wave + (unsigned)((slot * 2 + 1) * STAGE * 2 + j * 4096));
            }
            by += step_y;
            bx += step_x;
        }
        ++issued;
    };

    f32x4_t acc[4][4];
    f32x4_t bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bacc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
    Frag8 ones;
    ones.u[0] = ones.u[1] = ones.u[2] = ones.u[3] = CLV_ONE_PAIR;

    // per-lane LDS read addresses inside a stage tensor (bf16 element offsets): transpose-read block of rows
    // lg*4 + (lr>>2) .. , columns c0 + (lr&3)*4 .. with the pair swizzle; the second k-half is + 16 rows = + 2048 elements
    int ry[4], rx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lg * 4 + (lr >> 2);
        ry[i] = r * 128 + (((((wn + i * 16) >> 4) ^ (r & 7)) << 4) | ((lr & 3) << 2));
        rx[i] = r * 128 + (((((wk + i * 16) >> 4) ^ (r & 7)) << 4) | ((lr & 3) << 2));
    }
    auto tr = [](const bf16_t* p) {
        const v4s_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
        union { v4s_t v; uint2 u; } cv;
        cv.v = v;
        return cv.u;
    };
    auto compute = [&](const bf16_t* Ys, const bf16_t* Xs) {
        Frag8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i].u2[0] = tr(Ys + ry[i]);
            a[i].u2[1] = tr(Ys + ry[i] + 16 * 128);
            b[i].u2[0] = tr(Xs + rx[i]);
            b[i].u2[1] = tr(Xs + rx[i] + 16 * 128);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = (ACCUM && !RMW) ? mfma16(a[i], b[j], acc[i][j])   // D[n][k]: atomics coalesce along k across lanes
                                            : mfma16(b[j], a[i], acc[i][j]);  // swapped, D[k][n]: 4 consecutive k per lane
            if (do_bias) bacc[i] = (ACCUM && !RMW) ? mfma16(a[i], ones, bacc[i]) : mfma16(ones, a[i], bacc[i]);
        }
    };

    if (ws) {                                               // MFMA waves of a wave-specialised workgroup
        int sw = 0;
        for (; sw + RING <= nst; sw += RING) {
            __builtin_amdgcn_s_barrier(); compute(ring[0][0], ring[0][1]);
            __builtin_amdgcn_s_barrier(); compute(ring[1][0], ring[1][1]);
            __builtin_amdgcn_s_barrier(); compute(ring[2][0], ring[2][1]);
#if WG_RING == 4
            __builtin_amdgcn_s_barrier(); compute(ring[3][0], ring[3][1]);
#endif
        }
        int sl = 0;
        for (; sw < nst; ++sw) {
            __builtin_amdgcn_s_barrier();
            compute(ring[sl][0], ring[sl][1]);
            sl = sl == RING - 1 ? 0 : sl + 1;
        }
    } else {
#pragma unroll
    for (int d = 0; d < RING - 1; ++d) issue_slow(d);       // prologue (also correct for slices shorter than the ring)
    int st = 0;
    // ---- lean main loop: 4 stages per trip, all issues complete stages (issued + 3 < nfull), slots are constants
#define WG2_STAGE(SLOT)                                                                                     \
    wait_vm<(RING - 2) * 4>();                                                                              \
    __builtin_amdgcn_s_barrier();                                                                           \
    WG2_DMA(lds_wave, vy[0], vy[1], vx[0], vx[1], by, bx, ((SLOT + RING - 1) % RING) * 2 * STAGE * 2);    \
    by += step_y;                                                                                           \
    bx += step_x;                                                                                           \
    WG2_COMPUTE(ring[SLOT][0], ring[SLOT][1]);
#ifdef WG_ABL_NODMA
#define WG2_DMA(...)
#else
#define WG2_DMA dma4
#endif
#ifdef WG_ABL_NOCOMPUTE
#define WG2_COMPUTE(a, b)
#else
#define WG2_COMPUTE compute
#endif
    for (; issued + RING <= nfull; st += RING, issued += RING) {
        WG2_STAGE(0)
        WG2_STAGE(1)
        WG2_STAGE(2)
#if WG_RING == 4
        WG2_STAGE(3)
#endif
    }
#undef WG2_STAGE
    // ---- generic tail: at most 2 * RING - 1 stages (the last complete ones, the ragged one, the drain)
    int slot = 0;                                           // st is a multiple of RING here
    for (; st < nst; ++st) {
        if (issued - st == RING - 1 && issued <= nst) wait_vm<(RING - 2) * 4>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        issue_slow(slot == 0 ? RING - 1 : slot - 1);
        compute(ring[slot][0], ring[slot][1]);
        slot = slot == RING - 1 ? 0 : slot + 1;
    }
    wait_vm<0>();
    }
    float* pw = ACCUM ? out : out + (int64_t)split * ((int64_t)N * K + N);
    float* pb = ACCUM ? out_b : pw + (int64_t)N * K;
    if (ACCUM && RMW) {
        // one M-slice: this workgroup is the only writer of its dW elements in this launch (and the engine orders the
        // launches that share a parameter), so "+=" is a 16-byte load / add / store per lane in the swapped layout —
        // 2 x 4 B of traffic per element instead of a memory-side atomic each (~190 G/s: 12 us for a 768 x 3072 matrix).
        // overwrite: the step's FIRST gradient of this weight — dW is stale (the engine did not clear it): store, no load
        float ssq_q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn + i * 16 + lr;
            if (n >= N) continue;
            float4 old[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + wk + j * 16 + lg * 4;
                old[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < K && !(overwrite & 1)) old[j] = *reinterpret_cast<const float4*>(&pw[(int64_t)n * K + k]);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + wk + j * 16 + lg * 4;
                if (k < K) {
                    const float4 v = make_float4(old[j].x + acc[i][j][0], old[j].y + acc[i][j][1], old[j].z + acc[i][j][2],
                                                 old[j].w + acc[i][j][3]);
                    *reinterpret_cast<float4*>(&pw[(int64_t)n * K + k]) = v;
                    ssq_q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                }
            }
            if (do_bias && lg == 0) pb[n] += bacc[i][0];
        }
        if (ssq && (overwrite & 4)) ssq_commit(ssq, ssq_q);       // (wave-uniform condition)
    } else if (ACCUM) {
        // acc[i][j][r] = dW[n0 + wn + i*16 + lg*4 + r][k0 + wk + j*16 + lr]: this workgroup is the only writer of the
        // element; the no-return L2 atomic is a fire-and-forget "+=", one instruction = 4 rows x 64 contiguous bytes
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn + i * 16 + lg * 4 + r;
                if (n >= N) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + wk + j * 16 + lr;
                    if (k < K) atomicAdd(&pw[(int64_t)n * K + k], acc[i][j][r]);
                }
                if (do_bias && lr == 0) atomicAdd(&pb[n], bacc[i][r]);
            }
    } else {
        // swapped MFMA: acc[i][j][r] = dW[n0 + wn + i*16 + lr][k0 + wk + j*16 + lg*4 + r] — a lane owns 4 consecutive k:
        // one 16-byte store per (i, j), the 4 lanes of a row write 64 contiguous bytes (K % 8 == 0)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn + i * 16 + lr;
            if (n >= N) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + wk + j * 16 + lg * 4;
                if (k < K)
                    *reinterpret_cast<float4*>(&pw[(int64_t)n * K + k]) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            }
            if (do_bias && lg == 0) pb[n] = bacc[i][0];       // all-ones operand: every r holds the column sum
        }
    }
}

template <bool ACCUM, bool RMW = false>
__global__ void __launch_bounds__(WG_THREADS + 64, 2) wgrad_dma2_kernel(const bf16_t* __restrict__ dy,
                                                                   const bf16_t* __restrict__ x,
                                                                   float* __restrict__ out, float* __restrict__ out_b,
                                                                   int64_t M, int N, int K, int ldy, int ldx, int tiles,
                                                                   int tilesK, int nsplits, int64_t rows_per_split,
                                                                   int want_bias) {
    __shared__ __attribute__((aligned(1024))) bf16_t ring[RING][2][STAGE];   // [slot][dY | X]
    wgrad_dma2_body<ACCUM, RMW>(ring, blockIdx.x, dy, x, out, out_b, M, N, K, ldy, ldx, tiles, tilesK, nsplits, rows_per_split,
                           want_bias);
}

// Grouped launch: the weight gradients of a whole backward segment (they have no consumer before the optimizer) as ONE
// grid — workgroup b belongs to the problem whose [block_begin, next block_begin) holds b.  On their own these kernels
// are latency-bound (one workgroup per CU walking 50-100 stages behind 7 us of launch + prologue); together they keep
// every CU at two workgroups from different problems, need far fewer M-slices each (less partial traffic), and leave
// the dgrad chain of the backward pass uninterrupted.
constexpr int WG_GROUP_MAX = 80;      // (7.7 KB of kernel arguments: the AMDGPU kernarg segment is not bound to 4 KB)
struct WgProblem {
    const bf16_t* dy;
    const bf16_t* x;
    float* work;                                             // partials, or dW itself (in_place)
    float* db;                                               // in_place only
    int64_t M, rows_per_split;
    int N, K, ldy, ldx, tiles, tilesK, nsplits, want_bias, block_begin, xcd_rot, in_place;
    int overwrite;                                           // in_place only; bit 0: dW = (store), the step's first gradient of the weight; bit 2: add its sum of squares to the group's norm slots
};
struct WgGroup {
    WgProblem p[WG_GROUP_MAX];
    int n;
    float* ssq;                                              // norm slots (ssq_commit) or nullptr
};
__global__ void __launch_bounds__(WG_THREADS + 64, 2) wgrad_dma2_group_kernel(WgGroup grp) {
    __shared__ __attribute__((aligned(1024))) bf16_t ring[RING][2][STAGE];
    int idx = 0;
    for (int i = 1; i < grp.n; ++i)
        if ((int)blockIdx.x >= grp.p[i].block_begin) idx = i;
    const WgProblem& pr = grp.p[idx];
    if (pr.in_place)                                         // few-row problems: one slice, dW += in place
        wgrad_dma2_body<true, true>(ring, (int)blockIdx.x - pr.block_begin, pr.dy, pr.x, pr.work, pr.db, pr.M, pr.N, pr.K,
                                    pr.ldy, pr.ldx, pr.tiles, pr.tilesK, pr.nsplits, pr.rows_per_split, pr.want_bias,
                                    pr.xcd_rot, pr.overwrite, grp.ssq);
    else
        wgrad_dma2_body<false>(ring, (int)blockIdx.x - pr.block_begin, pr.dy, pr.x, pr.work, nullptr, pr.M, pr.N, pr.K,
                               pr.ldy, pr.ldx, pr.tiles, pr.tilesK, pr.nsplits, pr.rows_per_split, pr.want_bias,
                               pr.xcd_rot);
}

// ---------------------------------------------------------------------------------------------------
// Wide-tile version of the grouped kernel: a (128 PN) x (128 PK) output tile per workgroup of 4 PN PK waves (each wave
// still 64 x 64), stage = PN + PK panels of [32 rows][128 columns] (8 KiB each, same swizzle), ring of BIG_RING slots
// in dynamic LDS.  The 128 x 128 kernel re-reads every dY / X row once per tile that needs it; in a grouped launch the
// tiles of a slice drift apart on their XCD (L2 hit rate 40 %, 4.2 GB fetched for 2.2 GB of operands) and the launch
// runs at the fabric rate: identical time with the MFMAs removed (tools/probes/wgrad_group.py, build_wg_variants.sh).
// A 256 x 256 tile moves half the bytes per MFMA and has a quarter of the tiles to keep in step.
constexpr int BIG_RING = 4;
__device__ __forceinline__ void dma1(unsigned lds_byte, unsigned voff, const bf16_t* base) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[b]\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [lds] "s"(lds_byte), [v] "v"(voff), [b] "s"(base)
                 : "memory");
}

template <int PN, int PK, bool ACCUM>
__device__ __forceinline__ void wgrad_big_body(unsigned char* smem, const int bid, const bf16_t* __restrict__ dy,
                                               const bf16_t* __restrict__ x, float* __restrict__ out,
                                               float* __restrict__ out_b, int64_t M, int N, int K, int ldy, int ldx,
                                               int tiles, int tilesK, int nsplits, int64_t rows_per_split, int want_bias,
                                               int xcd_rot, int overwrite = 0, float* __restrict__ ssq = nullptr) {
    constexpr int P = PN + PK, W = 4 * PN * PK, IPW = 8 * P / W, GB = 2 * PK, R = BIG_RING;
    // LDS image: [panel][slot][8 KiB] — a wave's transpose reads of all slots then lie within the 64 KiB reach of the
    // ds immediate offset from ONE address register (slot-major order needed a register set per slot: spills)
    constexpr int SLOT = 8192;                               // bytes between the slots of one panel
    static_assert(8 * P % W == 0, "DMA instructions must divide evenly over the waves");
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, lr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                // scalar: piece tables, panel choices
    int split, tile;
    if (xcd_rot >= 0) {
        const int xcd = (bid - xcd_rot) & 7, xslot = bid >> 3;
        split = xcd + 8 * (xslot / tiles);
        tile = xslot % tiles;
    } else {
        split = bid / tiles;
        tile = bid - split * tiles;
    }
    if (split >= nsplits) return;
    const int tn = tile / tilesK, tk = tile - tn * tilesK;
    const int n0 = tn * 128 * PN, k0 = tk * 128 * PK;
    const int wa = wave / GB, wb = wave - wa * GB;           // this wave's 64 x 64 block of the tile
    const int wn = wa * 64, wk = wb * 64;
    const int64_t m_begin = (int64_t)split * rows_per_split;
    int64_t m_end = m_begin + rows_per_split;
    if (m_end > M) m_end = M;
    const bool do_bias = want_bias && tk == 0 && wb == 0;
    const int rows = (int)(m_end - m_begin);
    const int nst = (rows + SM - 1) / SM;
    const int nfull = rows / SM;

    // this wave's IPW pieces of a stage: piece q = wave * IPW + e -> panel q >> 3, rows (q & 7) * 4 .. + 4 of it
    unsigned voff[IPW], loff[IPW];
    int rowe[IPW];
    bool isx[IPW];
#pragma unroll
    for (int e = 0; e < IPW; ++e) {
        const int q = wave * IPW + e, panel = q >> 3, sub = q & 7;
        const int r = sub * 4 + (lane >> 4), c = ((lane & 15) ^ ((r & 7) << 1)) * 8;
        rowe[e] = r;
        isx[e] = panel >= PN;
        loff[e] = (unsigned)(panel * R * 8192 + sub * 1024);
        if (panel < PN) {
            int col = n0 + panel * 128 + c;
            col = col < N ? col : N - 8;
            voff[e] = (unsigned)((r * ldy + col) * 2);
        } else {
            int col = k0 + (panel - PN) * 128 + c;
            col = col < K ? col : K - 8;
            voff[e] = (unsigned)((r * ldx + col) * 2);
        }
    }
    const bf16_t* by = dy + m_begin * ldy;
    const bf16_t* bx = x + m_begin * ldx;
    const int64_t step_y = (int64_t)SM * ldy, step_x = (int64_t)SM * ldx;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem);
    int issued = 0;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero16);
    auto issue_slow = [&](int slot) {
        if (issued < nst) {
#pragma unroll
            for (int e = 0; e < IPW; ++e) {
                const bool in = issued * SM + rowe[e] < rows;
                const char* base = reinterpret_cast<const char*>(isx[e] ? bx : by);
                dma16(in ? reinterpret_cast<const bf16_t*>(base + voff[e]) : zero, lds0 + (unsigned)(slot * SLOT) + loff[e]);
            }
            by += step_y;
            bx += step_x;
        }
        ++issued;
    };

    f32x4_t acc[4][4];
    // bias on the VALU (registers are what limits this kernel at 4 waves per SIMD: an MFMA against all-ones costs 20):
    // a lane's dY fragment of block i is 8 rows of ONE column; the 4 lane groups meet at the end
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // transpose-read offsets (bf16 elements) inside this wave's dY / X panel, as in wgrad_dma2_body
    int ry[4], rx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = lg * 4 + (lr >> 2);
        ry[i] = (wa >> 1) * R * 4096 + r * 128 + ((((((wa & 1) * 64 + i * 16) >> 4) ^ (r & 7)) << 4) | ((lr & 3) << 2));
        rx[i] = (PN + (wb >> 1)) * R * 4096 + r * 128 + ((((((wb & 1) * 64 + i * 16) >> 4) ^ (r & 7)) << 4) | ((lr & 3) << 2));
    }
    auto tr = [](const bf16_t* p) {
        const v4s_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
        union { v4s_t v; uint2 u; } cv;
        cv.v = v;
        return cv.u;
    };
    auto compute = [&](const bf16_t* S) {
        Frag8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i].u2[0] = tr(S + ry[i]);
            a[i].u2[1] = tr(S + ry[i] + 16 * 128);
            b[i].u2[0] = tr(S + rx[i]);
            b[i].u2[1] = tr(S + rx[i] + 16 * 128);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(b[j], a[i], acc[i][j]);   // swapped: 4 consecutive k per lane
            if (do_bias) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    bsum[i] += half_lo(a[i].u[u]) + half_hi(a[i].u[u]);
            }
        }
    };

#pragma unroll
    for (int d = 0; d < R - 1; ++d) issue_slow(d);
    int st = 0;
    for (; issued + R <= nfull; st += R, issued += R) {
#pragma unroll
        for (int sl = 0; sl < R; ++sl) {
            wait_vm<(R - 2) * IPW>();
            __builtin_amdgcn_s_barrier();
#ifndef WG_ABL_NODMA
#pragma unroll
            for (int e = 0; e < IPW; ++e)
                dma1(lds0 + (unsigned)(((sl + R - 1) % R) * SLOT) + loff[e], voff[e], isx[e] ? bx : by);
#endif
            by += step_y;
            bx += step_x;
#ifndef WG_ABL_NOCOMPUTE
            compute(reinterpret_cast<const bf16_t*>(smem + sl * SLOT));
#endif
        }
    }
    int slot = 0;
    for (; st < nst; ++st) {
        if (issued - st == R - 1 && issued <= nst) wait_vm<(R - 2) * IPW>();
        else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        issue_slow(slot == 0 ? R - 1 : slot - 1);
        compute(reinterpret_cast<const bf16_t*>(smem + slot * SLOT));
        slot = slot == R - 1 ? 0 : slot + 1;
    }
    wait_vm<0>();
    float* pw = ACCUM ? out : out + (int64_t)split * ((int64_t)N * K + N);
    float* pb = ACCUM ? out_b : pw + (int64_t)N * K;
    float ssq_q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + wn + i * 16 + lr;
        if (n >= N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + wk + j * 16 + lg * 4;
            if (k >= K) continue;
            float4* dst = reinterpret_cast<float4*>(&pw[(int64_t)n * K + k]);
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (ACCUM && !(overwrite & 1)) {
                const float4 o = *dst;
                v = make_float4(v.x + o.x, v.y + o.y, v.z + o.z, v.w + o.w);
            }
            *dst = v;
            ssq_q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        if (do_bias) {
            const float bs = grp4_sum(bsum[i]);
            if (lg == 0) {
                if (ACCUM) pb[n] += bs;
                else pb[n] = bs;
            }
        }
    }
    if (ACCUM && ssq && (overwrite & 4)) ssq_commit(ssq, ssq_q);
}

template <int PN, int PK>
__global__ void __launch_bounds__(256 * PN * PK) wgrad_big_group_kernel(WgGroup grp) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char big_smem[];
    int idx = 0;
    for (int i = 1; i < grp.n; ++i)
        if ((int)blockIdx.x >= grp.p[i].block_begin) idx = i;
    const WgProblem& pr = grp.p[idx];
    if (pr.in_place)
        wgrad_big_body<PN, PK, true>(big_smem, (int)blockIdx.x - pr.block_begin, pr.dy, pr.x, pr.work, pr.db, pr.M, pr.N,
                                     pr.K, pr.ldy, pr.ldx, pr.tiles, pr.tilesK, pr.nsplits, pr.rows_per_split,
                                     pr.want_bias, pr.xcd_rot, pr.overwrite, grp.ssq);
    else
        wgrad_big_body<PN, PK, false>(big_smem, (int)blockIdx.x - pr.block_begin, pr.dy, pr.x, pr.work, nullptr, pr.M,
                                      pr.N, pr.K, pr.ldy, pr.ldx, pr.tiles, pr.tilesK, pr.nsplits, pr.rows_per_split,
                                      pr.want_bias, pr.xcd_rot);
}

// dw[e] += sum_s partial[s][e] (e < NK), db[e - NK] += ... (NK <= e < NK + N): a thread owns 4 consecutive e (16-byte
// loads, every wave load 1 KiB contiguous) for every SG-th slice, 4 loads in flight; the SG (1, 4 or 16) slice-lanes of
// an element group are 64 threads apart and meet in LDS.  SG grows as the matrix shrinks, so that the launch has enough
// threads (stage 0: 37 K elements x 86 slices).  (N*K and N are multiples of 8.)
// The first version (scalar loads, 16 split-lanes per element) ran at ~1 TB/s: 78 us for 75 MB of partials.
template <int SG>
__global__ void __launch_bounds__(256) fold_partials_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                            float* __restrict__ db, int64_t NK, int64_t E2,
                                                            int splits) {
    constexpr int EG = 256 / SG;                              // element groups (of 4) per block
    __shared__ float4 sh[SG > 1 ? 256 : 1];
    const int el = threadIdx.x % EG, sg = threadIdx.x / EG;
    const int64_t e = ((int64_t)blockIdx.x * EG + el) * 4;
    const int64_t Eeff = db ? E2 : NK;
    const bool in = e < Eeff;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
        int sp = sg;
        for (; sp + 3 * SG < splits; sp += 4 * SG) {
            const float4 v0 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 0 * SG) * E2 + e);
            const float4 v1 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 1 * SG) * E2 + e);
            const float4 v2 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 2 * SG) * E2 + e);
            const float4 v3 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 3 * SG) * E2 + e);
            a.x += (v0.x + v1.x) + (v2.x + v3.x);
            a.y += (v0.y + v1.y) + (v2.y + v3.y);
            a.z += (v0.z + v1.z) + (v2.z + v3.z);
            a.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; sp < splits; sp += SG) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)sp * E2 + e);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    if (SG > 1) {
        sh[threadIdx.x] = a;
        __syncthreads();
        if (sg != 0) return;
#pragma unroll
        for (int k = 1; k < SG; ++k) {
            const float4 v = sh[k * EG + el];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    if (!in) return;
    float* dst = e < NK ? dw + e : db + (e - NK);
    float4 o = *reinterpret_cast<const float4*>(dst);
    o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    *reinterpret_cast<float4*>(dst) = o;
}

// Batched fold: the partials of up to CLV_FOLD_MAX weight-gradient launches are folded into their dW / db in ONE launch
// (the table travels as a kernel argument) — a backward pass otherwise pays ~44 launch-bound 5-7 us fold kernels.
// Block = 256 threads = EG element groups (of 4 floats) x SG slice-lanes, SG = 1 << sg_shift per entry.
struct FoldTable {
    ClvFoldEntry e[CLV_FOLD_MAX];
    int n;
    float* ssq;                                              // norm slots (ssq_commit) or nullptr
};

__global__ void __launch_bounds__(256) fold_batch_kernel(FoldTable tab) {
    __shared__ float4 sh[256];
    int idx = 0;
    for (int i = 1; i < tab.n; ++i)
        if ((int)blockIdx.x >= tab.e[i].block_begin) idx = i;
    const ClvFoldEntry& en = tab.e[idx];
    const float* partial = static_cast<const float*>(en.partial);
    float* dw = static_cast<float*>(en.dw);
    float* db = static_cast<float*>(en.db);
    const int sg_shift = en.sg_shift, SG = 1 << sg_shift, EG = 256 >> sg_shift;
    const int el = threadIdx.x & (EG - 1), sg = threadIdx.x >> (8 - sg_shift);
    const int64_t e = ((int64_t)(blockIdx.x - en.block_begin) * EG + el) * 4;
    const int64_t NK = en.nk, E2 = en.e2, Eeff = db ? E2 : NK;
    const int splits = en.splits;
    const bool in = e < Eeff;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (in) {
        int sp = sg;
        for (; sp + 3 * SG < splits; sp += 4 * SG) {
            const float4 v0 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 0 * SG) * E2 + e);
            const float4 v1 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 1 * SG) * E2 + e);
            const float4 v2 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 2 * SG) * E2 + e);
            const float4 v3 = *reinterpret_cast<const float4*>(partial + (int64_t)(sp + 3 * SG) * E2 + e);
            a.x += (v0.x + v1.x) + (v2.x + v3.x);
            a.y += (v0.y + v1.y) + (v2.y + v3.y);
            a.z += (v0.z + v1.z) + (v2.z + v3.z);
            a.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; sp < splits; sp += SG) {
            const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)sp * E2 + e);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    }
    bool writer = in;
    if (SG > 1) {                                             // block-uniform
        sh[threadIdx.x] = a;
        __syncthreads();
        writer = in && sg == 0;
        if (writer)
            for (int k = 1; k < SG; ++k) {
                const float4 v = sh[k * EG + el];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
    }
    float q = 0.f;
    if (writer) {
        float* dst = e < NK ? dw + e : db + (e - NK);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        // overwrite (bit 0: dW, bit 1: db): the target is a fresh temporary, or the step's first gradient of the weight —
        // never read (nor zeroed)
        if (!(en.overwrite & (e < NK ? 1 : 2))) o = *reinterpret_cast<const float4*>(dst);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        *reinterpret_cast<float4*>(dst) = o;
        if (e < NK) q = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
    }
    // bit 2: the sum of squares of the dW this block wrote joins the gradient norm (one atomic per wave that wrote)
    if (tab.ssq && (en.overwrite & 4)) ssq_commit(tab.ssq, q);
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
    if (M <= 1024) return 1;                            // few rows: one slice, accumulate directly (no partials)
    // ~1 workgroup per CU: every extra slice costs N*K*4 bytes of partial traffic (store + fold), which for the
    // square-ish layers (stage 3, fusion: 144 tiles) is what the kernel's time consists of
    if (const char* f = getenv("CLV_WGRAD_SPLITS")) return atoi(f) < (M + 255) / 256 ? atoi(f) : (int)((M + 255) / 256);
    int64_t s = (256 + tiles - 1) / tiles;
    if (tiles >= 4 && tiles <= 40) s = (s + 7) / 8 * 8; // whole M-slices per XCD (see wgrad_dma_kernel), while that rounding is cheap
    const int64_t max_by_rows = (M + 255) / 256;        // >= 256 rows per slice
    if (s > max_by_rows) s = max_by_rows;
    if (s < 1) s = 1;
    return (int)s;
}

}  // namespace

// CLV_WGRAD_WS=1 (probe): 320 threads = four MFMA waves + one LDS-DMA producer wave (wgrad_dma2_body).  Measured and NOT
// the default: with two workgroups per CU the other workgroup already fills the issue bubbles — 11.89 vs 11.83 ms per step.
static unsigned wg_threads() {
    const char* v = getenv("CLV_WGRAD_WS");
    return (v && atoi(v) == 1) ? WG_THREADS + 64 : WG_THREADS;
}

extern "C" int64_t clv_linear_wgrad_work_floats(int64_t M, int32_t N, int32_t K) {
    const int tiles = ((N + TN - 1) / TN) * ((K + TK - 1) / TK);
    const int splits = pick_splits(M, tiles);
    return (int64_t)splits * ((int64_t)N * K + N);
}

extern "C" int clv_linear_wgrad(const void* dy, const void* x, float* dw, float* db, float* work, int64_t M,
                                int32_t N, int32_t K, int32_t ldy, int32_t ldx, const float* xmean,
                                const float* xrstd, int32_t stages, void* stream) {
    if (stages == 0) stages = 3;
    if ((xmean == nullptr) != (xrstd == nullptr)) return CLV_ERR_ARG;
    if (!dy || !x || !dw || !work || M <= 0 || N <= 0 || K <= 0 || (N & 7) || (K & 7) || (ldy & 7) || (ldx & 7))
        return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int tilesN = (N + TN - 1) / TN, tilesK = (K + TK - 1) / TK;
    const int tiles = tilesN * tilesK;
    const int splits = pick_splits(M, tiles);
    int64_t rows = (M + splits - 1) / splits;
    rows = (rows + TM - 1) / TM * TM;
    // up to 24 slices add their tiles into dW with no-return fp32 atomics (measured: cheaper than partials + fold;
    // with more slices the same-address contention at the memory-side atomic units costs more than the fold)
    // one slice: its tiles are added straight into dW (no-return fp32 atomics as fire-and-forget read-modify-writes);
    // several slices: fp32 partials + a streaming fold — atomics into dW from every slice measured 2x slower than the
    // whole rest of the kernel (190 G atomics/s: 25 us for the 18.9 MB of a stage-2 fc1, 50 us for a stage-3 fc1)
    const bool atomic_acc = getenv("CLV_WGRAD_ATOMIC") ? splits <= 24 : splits == 1;
    const bf16_t* dyp = (const bf16_t*)dy;
    const bf16_t* xp = (const bf16_t*)x;
    int rc = CLV_OK;
    if (stages & 1) {
        const int total = tiles * splits;
        const bool xcd_map = splits >= 8 && tiles >= 4;
        const int dma_grid = xcd_map ? 8 * tiles * ((splits + 7) / 8) : total;     // slots x 8 XCDs
        const int nsp = xcd_map ? splits : -1;
        if (xmean) {                       // standardise-on-load needs the register-staged kernel
            hipLaunchKernelGGL(wgrad_kernel, dim3(total), dim3(WG_THREADS), 0, st, dyp, xp, work, M, (int)N, (int)K,
                               (int)ldy, (int)ldx, tiles, tilesK, rows, db ? 1 : 0, xmean, xrstd);
        } else if (getenv("CLV_WGRAD_OLD")) {                                       // A/B switch: the first version
            if (atomic_acc)
                hipLaunchKernelGGL(wgrad_dma_kernel<true>, dim3(dma_grid), dim3(WG_THREADS), 0, st, dyp, xp, dw, db, M,
                                   (int)N, (int)K, (int)ldy, (int)ldx, tiles, tilesK, nsp, rows, db ? 1 : 0);
            else
                hipLaunchKernelGGL(wgrad_dma_kernel<false>, dim3(dma_grid), dim3(WG_THREADS), 0, st, dyp, xp, work, nullptr,
                                   M, (int)N, (int)K, (int)ldy, (int)ldx, tiles, tilesK, nsp, rows, db ? 1 : 0);
        } else if (atomic_acc && splits == 1 && !getenv("CLV_WGRAD_NORMW")) {   // one M-slice: dW += in place, no fold
            hipLaunchKernelGGL((wgrad_dma2_kernel<true, true>), dim3(dma_grid), dim3(wg_threads()), 0, st, dyp, xp, dw, db, M,
                               (int)N, (int)K, (int)ldy, (int)ldx, tiles, tilesK, nsp, rows, db ? 1 : 0);
        } else if (atomic_acc) {           // A/B switches: atomics straight into dW / db
            hipLaunchKernelGGL(wgrad_dma2_kernel<true>, dim3(dma_grid), dim3(wg_threads()), 0, st, dyp, xp, dw, db, M,
                               (int)N, (int)K, (int)ldy, (int)ldx, tiles, tilesK, nsp, rows, db ? 1 : 0);
        } else {
            hipLaunchKernelGGL(wgrad_dma2_kernel<false>, dim3(dma_grid), dim3(wg_threads()), 0, st, dyp, xp, work, nullptr,
                               M, (int)N, (int)K, (int)ldy, (int)ldx, tiles, tilesK, nsp, rows, db ? 1 : 0);
        }
        rc = clv_check_launch();
        if (rc) return rc;
    }
    if ((stages & 2) && !(atomic_acc && !xmean)) {
        const int64_t NK = (int64_t)N * K, E2 = NK + N;
        const int64_t Eeff = db ? E2 : NK;
        const int64_t groups = (Eeff + 3) / 4;               // threads needed with one slice-lane
        if (groups >= (1 << 17) || splits < 8)
            hipLaunchKernelGGL(fold_partials_kernel<1>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, work, dw,
                               db, NK, E2, splits);
        else if (groups >= (1 << 15) || splits < 32)
            hipLaunchKernelGGL(fold_partials_kernel<4>, dim3((unsigned)((groups + 63) / 64)), dim3(256), 0, st, work, dw,
                               db, NK, E2, splits);
        else
            hipLaunchKernelGGL(fold_partials_kernel<16>, dim3((unsigned)((groups + 15) / 16)), dim3(256), 0, st, work, dw,
                               db, NK, E2, splits);
        rc = clv_check_launch();
    }
    return rc;
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

extern "C" int clv_linear_wgrad_splits(int64_t M, int32_t N, int32_t K) {
    const int tiles = ((N + TN - 1) / TN) * ((K + TK - 1) / TK);
    return pick_splits(M, tiles);
}

extern "C" int clv_wgrad_fold_batch(const ClvFoldEntry* entries, int32_t n, void* stream) {
    return clv_wgrad_fold_batch_ss(entries, n, nullptr, stream);
}

extern "C" int clv_wgrad_fold_batch_ss(const ClvFoldEntry* entries, int32_t n, float* sumsq_slots, void* stream) {
    if (!entries || n <= 0 || n > CLV_FOLD_MAX) return CLV_ERR_ARG;
    static_assert(sizeof(ClvFoldEntry) == 56, "ClvFoldEntry layout is part of the ABI");
    FoldTable tab;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        ClvFoldEntry en = entries[i];
        if (!en.partial || !en.dw || en.splits <= 0 || (en.nk & 3) || ((en.e2 - en.nk) & 3)) return CLV_ERR_ARG;
        const int64_t groups = ((en.db ? en.e2 : en.nk) + 3) / 4;
        en.sg_shift = (groups >= (1 << 17) || en.splits < 8) ? 0 : (groups >= (1 << 15) || en.splits < 32) ? 2 : 4;
        en.block_begin = blocks;
        blocks += (int)((groups + (256 >> en.sg_shift) - 1) / (256 >> en.sg_shift));
        tab.e[i] = en;
    }
    tab.n = n;
    tab.ssq = sumsq_slots;
    hipLaunchKernelGGL(fold_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab);
    return clv_check_launch();
}

// Slices of one problem inside a grouped launch of n problems: enough workgroups in total (~8 per CU) rather than per
// problem, at least 256 rows per slice.
static int group_splits(int64_t M, int tiles, int n, int cls = 0) {
    int64_t target = (cls ? 1024 : 2048) / (n > 0 ? n : 1);
    if (target < (cls ? 24 : 64)) target = cls ? 24 : 64;
    if (cls && target > 48) target = 48;                     // few wide-tile problems: long slices beat more partials (-0.05 ms)
    static const int env_big = getenv("CLV_WGRAD_BIG_TARGET") ? atoi(getenv("CLV_WGRAD_BIG_TARGET")) : 0;
    if (cls && env_big > 0) target = env_big;
    static const int env_target = getenv("CLV_WGRAD_GROUP_TARGET") ? atoi(getenv("CLV_WGRAD_GROUP_TARGET")) : 0;
    static const int env_rows = getenv("CLV_WGRAD_GROUP_ROWS") ? atoi(getenv("CLV_WGRAD_GROUP_ROWS")) : 0;
    if (env_target > 0) target = env_target;
    int64_t s = (target + tiles - 1) / tiles;
    if (env_rows > 0) s = (M + env_rows - 1) / env_rows;
    const int64_t max_by_rows = (M + 255) / 256;
    if (s > max_by_rows) s = max_by_rows;
    if (s < 1) s = 1;
    return (int)s;
}

// Tile class of a grouped problem: 1 = 256 x 256 tiles (wgrad_big_group_kernel<2, 2>: half the staged bytes per MFMA)
// for outputs that such tiles cover without overhang; everything else keeps 128 x 128 tiles — with overhang the wide
// tiles spend 30-80 % more MFMA / LDS time on clamped columns, and the compute side alone (479 us of the 760 us launch)
// then exceeds what the DMA side saves (measured: 813 us with every shape that staged >= 20 % fewer bytes on wide tiles).
static int wg_class(int N, int K, int64_t M = 1 << 20) {
    static const int big = getenv("CLV_WGRAD_BIG") ? atoi(getenv("CLV_WGRAD_BIG")) : 1;
    static const int small_m = getenv("CLV_WGRAD_BIG_SMALLM") ? atoi(getenv("CLV_WGRAD_BIG_SMALLM")) : 0;   // few-row problems: 128 x 128 tiles (A/B: 0.05 ms better)
    static const int rect = getenv("CLV_WGRAD_RECT") ? atoi(getenv("CLV_WGRAD_RECT")) : 0;   // probe: 128 x 256 / 256 x 128 classes (+0.25 ms on the step: two more launches, one 8-wave workgroup per CU)
    if (M <= 1024 && !small_m) return 0;
    if (!big) return 0;
    if (N % 256 == 0 && K % 256 == 0) return 1;
    if (rect && N % 128 == 0 && K % 256 == 0) return 2;      // 128 x 256 tiles, 8 waves
    if (rect && N % 256 == 0 && K % 128 == 0) return 3;      // 256 x 128
    return 0;
}
constexpr int WG_CLASSES = 4;                               // the fixed-tile classes
static const int wg_tn[WG_CLASSES] = {128, 256, 128, 256}, wg_tk[WG_CLASSES] = {128, 256, 256, 128};
// One M-slice accumulated straight into dW / db (no partials, no fold): few rows, or an output so large that every extra
// slice costs more partial traffic (write + fold read of N x K floats) than it saves in workgroup length.
static bool wg_in_place(int64_t M, int N, int K) {
    return M <= 1024 || (N % 256 == 0 && K % 256 == 0 && (int64_t)N * K > (1 << 20));
}
extern "C" int clv_linear_wgrad_in_place(int64_t M, int32_t N, int32_t K) { return wg_in_place(M, N, K) ? 1 : 0; }
extern "C" int clv_linear_wgrad_class(int64_t M, int32_t N, int32_t K) { return wg_class(N, K, M); }
static int wg_tiles(int N, int K, int cls) {
    return ((N + wg_tn[cls] - 1) / wg_tn[cls]) * ((K + wg_tk[cls] - 1) / wg_tk[cls]);
}

extern "C" int clv_linear_wgrad_batch_plan(ClvWgradEntry* entries, int32_t n) {
    if (!entries || n <= 0 || n > WG_GROUP_MAX) return CLV_ERR_ARG;
    int ncls[WG_CLASSES] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const ClvWgradEntry& e = entries[i];
        if (e.M <= 0 || e.N <= 0 || e.K <= 0 || (e.N & 7) || (e.K & 7)) return CLV_ERR_ARG;
        ++ncls[wg_class(e.N, e.K, e.M)];
    }
    for (int i = 0; i < n; ++i) {
        ClvWgradEntry& e = entries[i];
        const int cls = wg_class(e.N, e.K, e.M);
        if (wg_in_place(e.M, e.N, e.K)) {
            e.splits = 1;
            e.work_floats = 0;
            continue;
        }
        e.splits = group_splits(e.M, wg_tiles(e.N, e.K, cls), ncls[cls], cls);
        e.work_floats = (int64_t)e.splits * ((int64_t)e.N * e.K + e.N);
    }
    return CLV_OK;
}

extern "C" int clv_linear_wgrad_batch(const ClvWgradEntry* entries, int32_t n, void* stream) {
    return clv_linear_wgrad_batch_ss(entries, n, nullptr, stream);
}

extern "C" int clv_linear_wgrad_batch_ss(const ClvWgradEntry* entries, int32_t n, float* sumsq_slots, void* stream) {
    if (!entries || n <= 0 || n > WG_GROUP_MAX) return CLV_ERR_ARG;
    static_assert(sizeof(WgGroup) <= 8000, "kernel-argument budget");
    // Longest workgroups first: a problem's workgroups walk M / splits rows each (1 500 ... 9 000 inside one launch), blocks
    // are dispatched in index order, and the problems arrive in backward order — stage 0 with the longest slices LAST, where
    // its ~260 workgroups used to run on alone after everything else had drained.
    int order[WG_GROUP_MAX];
    for (int i = 0; i < n; ++i) order[i] = i;
    static const bool lpt = !getenv("CLV_WGRAD_LPT") || atoi(getenv("CLV_WGRAD_LPT")) != 0;
    if (lpt)
        std::stable_sort(order, order + n, [&](int a, int b) {
            return entries[a].M / (entries[a].splits > 0 ? entries[a].splits : 1) >
                   entries[b].M / (entries[b].splits > 0 ? entries[b].splits : 1);
        });
    for (int cls = 0; cls < WG_CLASSES; ++cls) {
        WgGroup grp;
        int blocks = 0, rot = 0, cnt = 0;
        for (int oi = 0; oi < n; ++oi) {
            const int i = order[oi];
            const ClvWgradEntry& e = entries[i];
            const bool in_place = e.work_floats == 0;
            if (!e.dy || !e.x || e.splits <= 0 || (e.ldy & 7) || (e.ldx & 7) || e.N <= 0 || e.K <= 0) return CLV_ERR_ARG;
            if (in_place ? (!e.dw || e.splits != 1 || (e.want_bias && !e.db)) : !e.work) return CLV_ERR_ARG;
            if (wg_class(e.N, e.K, e.M) != cls) continue;
            WgProblem& p = grp.p[cnt++];
            const int tilesN = (e.N + wg_tn[cls] - 1) / wg_tn[cls], tilesK = (e.K + wg_tk[cls] - 1) / wg_tk[cls];
            p.dy = (const bf16_t*)e.dy;
            p.x = (const bf16_t*)e.x;
            p.work = in_place ? e.dw : (float*)e.work;
            p.db = in_place ? e.db : nullptr;
            p.in_place = in_place;
            p.overwrite = in_place ? (e.overwrite & 5) : 0;
            p.M = e.M;
            p.N = e.N; p.K = e.K; p.ldy = e.ldy; p.ldx = e.ldx;
            p.tiles = tilesN * tilesK;
            p.tilesK = tilesK;
            int64_t rows = (e.M + e.splits - 1) / e.splits;
            p.rows_per_split = (rows + TM - 1) / TM * TM;
            // The tiles of one M-slice always share an XCD (consecutive slots of it), so the dY / X rows they all read
            // come from that L2 — without this a grouped launch ran at the Infinity-Cache rate (6.9 TB/s of LDS-DMA
            // traffic).  Slices beyond nsplits exit at once; the XCD a problem's first slice uses rotates with the
            // slices placed so far.
            const bool xcd_map = p.tiles >= 2 && !in_place;  // one slice: its tiles over all XCDs
            p.nsplits = e.splits;
            p.want_bias = e.want_bias;
            p.block_begin = blocks;
            p.xcd_rot = xcd_map ? (rot & 7) : -1;            // -1: plain (slice-major) block order
            if (xcd_map) rot += e.splits;
            int g = xcd_map ? 8 * p.tiles * ((e.splits + 7) / 8) : p.tiles * e.splits;
            blocks += (g + 7) / 8 * 8;                       // keep every problem's block ids aligned to the 8 XCDs
        }
        if (!cnt) continue;
        grp.n = cnt;
        grp.ssq = sumsq_slots;
        if (cls == 0) {
            hipLaunchKernelGGL(wgrad_dma2_group_kernel, dim3((unsigned)blocks), dim3(wg_threads()), 0, (hipStream_t)stream,
                               grp);
        } else if (cls == 1) {
            constexpr int LDS = BIG_RING * 4 * 8192;
            static const bool attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_big_group_kernel<2, 2>),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
            (void)attr;
            hipLaunchKernelGGL((wgrad_big_group_kernel<2, 2>), dim3((unsigned)blocks), dim3(1024), LDS,
                               (hipStream_t)stream, grp);
        } else {
            constexpr int LDS = BIG_RING * 3 * 8192;
            static const bool attr12 = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_big_group_kernel<1, 2>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
            static const bool attr21 = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_big_group_kernel<2, 1>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS) == hipSuccess;
            (void)attr12; (void)attr21;
            if (cls == 2)
                hipLaunchKernelGGL((wgrad_big_group_kernel<1, 2>), dim3((unsigned)blocks), dim3(512), LDS,
                                   (hipStream_t)stream, grp);
            else
                hipLaunchKernelGGL((wgrad_big_group_kernel<2, 1>), dim3((unsigned)blocks), dim3(512), LDS,
                                   (hipStream_t)stream, grp);
        }
        const int rc = clv_check_launch();
        if (rc) return rc;
    }
    return CLV_OK;
}
