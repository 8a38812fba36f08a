// LDS-tiled bf16 MFMA GEMM with fused epilogues for the mid-size Linear layers of the step (Swin stages 1-3, the
// fusion encoder; M = 3 000 .. 50 000 rows, N and K = 192 .. 3072):
//        C[M][N] = A[M][K] * B[N][K]^T  (+ bias)  (GELU | * GELU'(aux))
// Both operands are K-contiguous ("NT"), which is what every GEMM of a Linear layer is once the weight is also kept
// transposed:   forward  y  = x  W^T        A = x  [M][K],   B = W   [N][K]      (Mlp.fc1/fc2, qkv, proj:
//               dgrad    dx = dy W          A = dy [M][N],   B = W^T [K][N]       swin_transformer_3d.py:262-268,376,398)
// (the engine keeps a bf16 W^T next to the bf16 W shadow, refreshed once per step — clv_transpose_batch).
// Epilogues: bias; bias + erf-GELU with the pre-activation kept (fc1 forward — replaces GEMM + GELU kernel);
// multiply by GELU'(pre) (fc2 input gradient — replaces GEMM + GELU-backward kernel).
//
// Structure (gfx950): 256 x 128 (or 128 x 128) output tile per 512-thread workgroup (8 waves, 64 x 64 / 64 x 32 each,
// 16x16x32 bf16 MFMA with swapped operands), 64-deep K stages brought global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring that fills the LDS, counted vmcnt waits
// + one raw s_barrier per stage; ONE persistent workgroup per CU whose tiles form one flat pipeline (see the kernel).  LDS rows are 128 B
// (64 bf16: full cache lines — 64-byte rows fetch at half the rate) with the 16-byte chunk index XORed with (row & 7)
// on the SOURCE side (the DMA writes lane-linear), which makes every ds_read_b128 of an MFMA operand bank-conflict free.  The epilogue runs on the accumulators and stores 16 bytes per lane and instruction (no
// LDS round trip); the bias row sits in LDS.  The tiles of one row block run on one XCD (blockIdx % 8), so A rows are
// fetched from HBM once per XCD L2; B (the weight) is shared by all.
#include "common.hpp"
#include "../../include/clover_hip.h"
#include <stdlib.h>
#include <string.h>

namespace {

#ifdef GN_TRACE          // tools/probes/gemm_lab.cpp: per-workgroup cycle sums of the main loop's phases (wave 0)
__device__ unsigned long long* gn_trace_buf = nullptr;
#define GN_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define GN_T(var)
#endif

constexpr int GN_BK = 64;                                      // bf16 elements per stage row = 128 B (a full L2 line:
                                                               // 64-byte rows fetch at half the rate, tools/probes/dma_rate.cpp)
constexpr int GN_MAX_BIAS = 3072;                              // bias row kept in LDS (as bf16)

__device__ __forceinline__ void gn_dma16(const bf16_t* src, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_byte)
                 : "memory");
}
// One DMA piece (1 KiB) on its own: base (SGPR pair) + per-lane byte offset -> LDS byte address lds.
__device__ __forceinline__ void gn_piece(unsigned lds, unsigned voff, const bf16_t* base) {
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v], %[b]\n\ts_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [lds] "s"(lds), [v] "v"(voff), [b] "s"(base)
                 : "memory");
}
template <int N_>
__device__ __forceinline__ void gn_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// One operand stage: ROWS rows x 64 k.  A DMA piece is 1 KiB = 8 LDS rows x 128 B, written lane-linear: lane l lands at
// LDS row l >> 3, chunk position l & 7, and fetches chunk (l & 7) ^ (row & 7) of the GLOBAL row that LDS row holds
// (with this XOR the 16 rows x 16 B of an MFMA operand read cover the 16 slots of a 256-byte bank row once per
// ds_read_b128 lane group).  ROWS / 64 pieces per wave.  Rows past the edge are CLAMPED to the last valid row: they feed
// only accumulators that are never stored.  PERM_TN > 0 (the B operand): the LDS image is row-permuted so that the
// swapped-operand MFMAs leave each lane with 8 CONSECUTIVE output columns per accumulator pair and the 4 lanes of a row
// with 64 contiguous bytes per store instruction: LDS row j*16 + x of a wave's WN-row block holds global row
// (j >> 1) * 32 + (x >> 2) * 8 + (j & 1) * 4 + (x & 3) — the per-lane source address makes any row permutation free.
template <int NPIECE>
__device__ __forceinline__ void gn_dma_block(unsigned lds, const unsigned (&v)[NPIECE], const bf16_t* base);
template <>
__device__ __forceinline__ void gn_dma_block<4>(unsigned lds, const unsigned (&v)[4], const bf16_t* base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [lds] "s"(lds), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [b] "s"(base)
        : "memory", "scc");
}
template <>
__device__ __forceinline__ void gn_dma_block<1>(unsigned lds, const unsigned (&v)[1], const bf16_t* base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [lds] "s"(lds), [v0] "v"(v[0]), [b] "s"(base)
        : "memory", "scc");
}
template <>
__device__ __forceinline__ void gn_dma_block<6>(unsigned lds, const unsigned (&v)[6], const bf16_t* base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v4], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x1400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v5], %[b]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [lds] "s"(lds), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [b] "s"(base)
        : "memory", "scc");
}
template <>
__device__ __forceinline__ void gn_dma_block<3>(unsigned lds, const unsigned (&v)[3], const bf16_t* base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [lds] "s"(lds), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [b] "s"(base)
        : "memory", "scc");
}
template <>
__device__ __forceinline__ void gn_dma_block<2>(unsigned lds, const unsigned (&v)[2], const bf16_t* base) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %[keep], m0\n\t"
        "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
        "s_add_u32 m0, %[lds], 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep)
        : [lds] "s"(lds), [v0] "v"(v[0]), [v1] "v"(v[1]), [b] "s"(base)
        : "memory", "scc");
}


// The DMA pieces of a wave go out in the SGPR-base + 32-bit-VGPR-offset form, one asm block per operand and stage (M0 saved /
// restored once): the per-lane offsets are fixed for a tile, a stage advance is ONE scalar add on the base.  The first version
// kept a 64-bit per-lane pointer per piece (readfirstlane + two M0 moves + a 64-bit VALU add per piece): the pieces' issue
// cost — not their latency — paced the few-tile classes (6 pieces per 16 MFMAs and wave at 64 x 128; a ring of 6 changed
// nothing, tools/probes/gemm_tiles.py).
template <int ROWS, int WAVES>
struct GnOper {
    static constexpr int PIECES = ROWS / (8 * WAVES);
    unsigned voff[PIECES];              // bytes from `base`, per lane
    const bf16_t* base;                 // wave-uniform: first row of the tile, current k offset
};

template <int ROWS, int WAVES, int PERM_TN>
__device__ __forceinline__ void gn_init(GnOper<ROWS, WAVES>& o, const bf16_t* base, int64_t row0, int64_t nrows, int64_t ld, int wave,
                                        int lane) {
    const int64_t r0 = row0 < nrows ? row0 : nrows - 1;       // a tile past the edge (never stored) reads the last row
    o.base = base + r0 * ld;
#pragma unroll
    for (int j = 0; j < GnOper<ROWS, WAVES>::PIECES; ++j) {
        const int r = (wave * GnOper<ROWS, WAVES>::PIECES + j) * 8 + (lane >> 3);      // LDS row
        const int c = (lane & 7) ^ (r & 7);
        int g = r;                                                               // global row held by LDS row r
        if (PERM_TN > 0) {
            constexpr int WN = PERM_TN * 16;
            const int h = r / WN, p = r % WN, jj = p >> 4, x = p & 15;
            g = h * WN + (jj >> 1) * 32 + (x >> 2) * 8 + (jj & 1) * 4 + (x & 3);
        }
        int64_t gr = r0 + g;
        gr = gr < nrows ? gr : nrows - 1;
        o.voff[j] = (unsigned)(((gr - r0) * ld + c * 8) * 2);
    }
}

template <int ROWS, int WAVES>
__device__ __forceinline__ void gn_issue(GnOper<ROWS, WAVES>& o, unsigned lds_oper_base, int wave) {
#ifndef GN_ABL_NODMA
    gn_dma_block<GnOper<ROWS, WAVES>::PIECES>(
        __builtin_amdgcn_readfirstlane(lds_oper_base + (unsigned)(wave * GnOper<ROWS, WAVES>::PIECES * 1024)), o.voff, o.base);
#endif
    o.base += GN_BK;
}

__device__ __forceinline__ Frag8 gn_frag(const unsigned char* oper, int r, int kc) {
    Frag8 f;
    f.u4 = *reinterpret_cast<const uint4*>(oper + r * 128 + ((kc ^ (r & 7)) << 4));
    return f;
}

enum { GN_EPI_NONE = CLV_GEMM_EPI_NONE, GN_EPI_BIAS = CLV_GEMM_EPI_BIAS, GN_EPI_GELU = CLV_GEMM_EPI_BIAS_GELU,
       GN_EPI_DGELU = CLV_GEMM_EPI_DGELU, GN_EPI_GELUD = CLV_GEMM_EPI_BIAS_GELU_D, GN_EPI_MUL = CLV_GEMM_EPI_MUL,
       GN_EPI_PARTIAL = 6 };        // split-K slice: fp32 partial sums to the work slab, the epilogue runs in the reduce kernel

__device__ __forceinline__ float gn_lo(uint32_t u) { return half_lo(u); }
__device__ __forceinline__ float gn_hi(uint32_t u) { return half_hi(u); }

// ONE persistent 8-wave workgroup per CU (waves as WAVES_M x WAVES_N, two per SIMD); workgroup w lives on XCD w & 7
// (dispatch order) and walks the tiles of the row blocks mblk = xcd (mod 8) in (mblk, tn) order with stride gridDim / 8,
// so the tiles that share A rows run on one XCD's L2.  The 64-deep stages of ALL its tiles form one flat software
// pipeline through a ring of R slots that fills the LDS: while stage g is multiplied, stages g+1 .. g+R-1 — the next
// tile's first stages included — are in flight (~100 KB per CU: LDS-DMA round trips measure 1 100 cycles from L2 and
// 2 700 from the Infinity Cache), with counted vmcnt waits and one raw s_barrier per stage (LDS-DMA requests stay in
// flight across it).  The epilogue runs straight on the accumulators; its stores are younger than the stages already in
// flight, which the wait counts of the next R-1 stages account for (gfx9 returns loads and stores in issue order on
// vmcnt), so the stores drain under the next tile's first stages instead of stalling the pipeline.
//
// FP8 = true (clv_gemm_nt_fp8): the operands are OCP e4m3 bytes.  The staging side is unchanged — a stage row is still 128
// bytes, now 128 k-elements — so a, b, K, lda, ldb arrive in 2-byte units (K / 2 ...); a stage is ONE k-step of the
// K = 128 matrix instruction (v_mfma_f32_16x16x128_f8f6f4, twice the bf16 rate): a lane's operand is the 32 bytes at
// chunks 2 lg, 2 lg + 1 of its row (the same assignment for A and B, so the order of k inside a stage is immaterial).
// The epilogue applies the per-row scales of the two quantised operands: acc * sa[m] * sb[n] (+ bias ...).
typedef int gn_i32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4_t gn_mfma_fp8(const Frag8 (&x)[2], const Frag8 (&y)[2], f32x4_t acc) {
    const gn_i32x8_t xv = {(int)x[0].u[0], (int)x[0].u[1], (int)x[0].u[2], (int)x[0].u[3],
                           (int)x[1].u[0], (int)x[1].u[1], (int)x[1].u[2], (int)x[1].u[3]};
    const gn_i32x8_t yv = {(int)y[0].u[0], (int)y[0].u[1], (int)y[0].u[2], (int)y[0].u[3],
                           (int)y[1].u[0], (int)y[1].u[1], (int)y[1].u[2], (int)y[1].u[3]};
    // cbsz = blgp = 0: both operands e4m3; block scales 2^0 (E8M0 127) — the tensors carry per-row fp32 scales instead
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(xv, yv, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int R, int EPI, bool FP8 = false>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N, 2) gemm_nt_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                                const float* __restrict__ bias, const bf16_t* __restrict__ aux,
                                                                bf16_t* __restrict__ c, bf16_t* __restrict__ c2, int64_t M, int N,
                                                                int K, int64_t lda, int64_t ldb, int64_t ldc, int tilesN,
                                                                int nmblk, const float* __restrict__ sa = nullptr,
                                                                const float* __restrict__ sb = nullptr, int pc = 1,
                                                                int splitk = 1, int rot_on = 0,
                                                                float* __restrict__ partial = nullptr) {
    constexpr int WAVES = WAVES_M * WAVES_N, GN_THREADS = 64 * WAVES;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int P = GnOper<BM, WAVES>::PIECES + GnOper<BN, WAVES>::PIECES;     // DMA pieces per wave and stage
    constexpr int S = TM * (TN / 2) * ((EPI == GN_EPI_GELU || EPI == GN_EPI_GELUD || EPI == GN_EPI_PARTIAL) ? 2 : 1);      // store instructions per wave and epilogue
    constexpr bool HAS_BIAS = EPI == GN_EPI_BIAS || EPI == GN_EPI_GELU || EPI == GN_EPI_GELUD;
    static_assert((R - 2) * P + S <= 63, "vmcnt immediate");
    static_assert(TN % 2 == 0 && R >= 2, "tile shape");
    __shared__ __attribute__((aligned(1024))) unsigned char ring[R * STAGE + (HAS_BIAS ? GN_MAX_BIAS * 2 : 0)];
    bf16_t* bias_s = reinterpret_cast<bf16_t*>(ring + R * STAGE);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    // The tile grid is split over the 8 XCDs as (8 / pc) x pc: XCD x = xr * pc + xc owns the row blocks = xr (mod 8 / pc) and
    // the column tiles = xc (mod pc), so one L2 serves A_bytes * pc / 8 + B_bytes / pc.  pc = 1 (rows only) suits the
    // token-parallel layers (a small weight, every XCD reads all of it); for M ~ 3 000 rows against a 3-5 MB weight the
    // rows-only split puts the WHOLE weight + an eighth of A (> 4 MB) through every L2 and the DMA runs at the
    // Infinity-Cache rate — the host picks the pc with the smallest per-XCD working set.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int pr = 8 / pc, xr = xcd / pc, xc = xcd - xr * pc;
    const int my_mblks = (nmblk - xr + pr - 1) / pr;
    const int my_tns = (tilesN - xc + pc - 1) / pc;
    const int my_tiles = my_mblks * my_tns;
    // A work UNIT is a (tile, K slice) pair: splitk > 1 cuts the contraction of every tile into splitk slices of whole
    // stages that run as units of their own (consecutive slots) and leave fp32 partial sums for the reduce kernel — the
    // few-tile long-contraction layers (fc2 / fc1-dgrad of Swin stage 3, the fusion encoder, the text tower: 24-174 tiles
    // of 128 x 128 against 512 workgroup slots, 48 stages each) fill the chip that way.
    const int my_units = my_tiles * splitk;
    if (slot >= my_units || my_tns <= 0) return;
    auto tile_m = [&](int t) { return xr + pr * (t / my_tns); };      // row block / column tile of this XCD's t-th tile
    auto tile_n = [&](int t) { return xc + pc * (t % my_tns); };
    const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
    const unsigned ring_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&ring[0]);
    const int nst_all = K / GN_BK;
    // stages [kb, kb + cnt) of unit u, and the stage its K walk starts at: with rot_on the tiles of an XCD start their
    // walks at different stages (a Latin square over the XCD's row blocks x column tiles) and wrap around.  All the
    // workgroups of a launch run in step, so without the rotation every A / B line of stage g is requested by its 6-9
    // sharers AT ONCE and each of them waits for the miss (Infinity Cache / HBM: ~2 700 cycles against ~1 100 from L2);
    // rotated, a line's first requester takes the miss and the others find it in L2 later.  fp32 sums over the stages in
    // a different order per tile — deterministic for a given shape.
    auto unit_span = [&](int u, int& kb, int& cnt, int& rot) {
        const int t = u / splitk, z = u - t * splitk;
        kb = (nst_all * z) / splitk;
        cnt = (nst_all * (z + 1)) / splitk - kb;
        rot = rot_on ? ((t / my_tns) + (t % my_tns) + z) % cnt : 0;
        return t;
    };

    GnOper<BM, WAVES> A;
    GnOper<BN, WAVES> B;
    int qn = slot;                                            // unit whose stages are being issued
    int issued = 0;                                           // its stages issued so far
    int icnt = 0, ikpos = 0;                                  // its stage count; position of the next issue inside its span
    unsigned ibase = ring_base;                               // ring slot (LDS byte address) the next issue goes to
    int inflight = 0;                                         // stages issued and not yet consumed
    auto init_unit = [&](int u) {
        int kb, rot;
        const int t = unit_span(u, kb, icnt, rot);
        ikpos = rot;
        gn_init<BM, WAVES, 0>(A, a + (int64_t)(kb + rot) * GN_BK, (int64_t)tile_m(t) * BM, M, lda, wave, lane);
        gn_init<BN, WAVES, TN>(B, b + (int64_t)(kb + rot) * GN_BK, tile_n(t) * BN, N, ldb, wave, lane);
    };
    init_unit(qn);
    auto issue_next = [&]() {                                 // one stage of the flat (unit, stage) sequence, if any is left
        if (qn >= my_units) return;
        gn_issue<BM, WAVES>(A, ibase, wave);
        gn_issue<BN, WAVES>(B, ibase + A_BYTES, wave);
        ibase = ibase == ring_base + (R - 1) * STAGE ? ring_base : ibase + STAGE;
        ++inflight;
        if (++ikpos == icnt) {                                // wrap of a rotated walk: back to the span's first stage
            ikpos = 0;
            A.base -= (int64_t)icnt * GN_BK;
            B.base -= (int64_t)icnt * GN_BK;
        }
        if (++issued == icnt) {
            issued = 0;
            qn += nslot;
            if (qn < my_units) init_unit(qn);
        }
    };
#pragma unroll
    for (int d = 0; d < R - 1; ++d) issue_next();
    const bool bias_lds = N <= GN_MAX_BIAS;                   // wider outputs (the MLM decoder) read their bias from global memory
    if (HAS_BIAS && bias_lds) {
        // whole bias row -> LDS (bf16, as the GEMM's bf16 bias operand was) UNDER the first stages' DMA round trip (it
        // used to run, with a barrier, before the first DMA went out: ~1 us of serial latency per launch).  The first
        // stage's barrier below publishes it: its LDS writes are waited for here.
        for (int i = tid * 4; i < N; i += GN_THREADS * 4) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + i);
            *reinterpret_cast<uint2*>(bias_s + i) = make_uint2(pack2bf(bv.x, bv.y), pack2bf(bv.z, bv.w));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0)
    }
    int cslot = 0;                                            // ring slot of the stage being multiplied
    int post_epi = 0;                                         // stage tops whose wait must also leave S stores in flight
#ifdef GN_TRACE
    unsigned long long tr_wait = 0, tr_bar = 0, tr_issue = 0, tr_comp = 0, tr_epi = 0, tr_stages = 0;
    const unsigned long long tr_start = __builtin_amdgcn_s_memtime();
#endif

    for (int q = slot; q < my_units; q += nslot) {
        int kb_q, nst, rot_q;
        const int tq = unit_span(q, kb_q, nst, rot_q);
        const int zq = q - tq * splitk;
        f32x4_t acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        for (int st = 0; st < nst; ++st) {
            // the oldest stage in flight must have landed; a full pipeline leaves R-2 younger stages (and, right after an
            // epilogue, its S stores) outstanding
            GN_T(t0);
            if (inflight == R - 1) {
                if (post_epi > 0) { gn_wait_vm<(R - 2) * P + S>(); --post_epi; }
                else gn_wait_vm<(R - 2) * P>();
            } else {
                gn_wait_vm<0>();
                post_epi = 0;
            }
            GN_T(t1);
            __builtin_amdgcn_s_barrier();                     // everyone's pieces landed; slot of stage g-1 is drained
            GN_T(t2);
            --inflight;
            issue_next();
            GN_T(t3);
            const unsigned char* As = ring + cslot * STAGE;
            const unsigned char* Bs = As + A_BYTES;
            cslot = cslot == R - 1 ? 0 : cslot + 1;
            if constexpr (FP8) {
                Frag8 fa[TM][2], fb[TN][2];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    fb[j][0] = gn_frag(Bs, wn + j * 16 + lr, 2 * lg);
                    fb[j][1] = gn_frag(Bs, wn + j * 16 + lr, 2 * lg + 1);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    fa[i][0] = gn_frag(As, wm + i * 16 + lr, 2 * lg);
                    fa[i][1] = gn_frag(As, wm + i * 16 + lr, 2 * lg + 1);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = gn_mfma_fp8(fb[j], fa[i], acc[i][j]);   // swapped: D[n][m]
            } else
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                Frag8 fa[TM], fb[TN];
#ifdef GN_ABL_NOLDS
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i].u4 = make_uint4(0x3c003c00u + st, 0x3c003c00u, 0x3c003c00u + i, 0x3c003c00u);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j].u4 = make_uint4(0x3c003c00u + j, 0x3c003c00u, 0x3c003c00u + lane, 0x3c003c00u);
#else
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = gn_frag(Bs, wn + j * 16 + lr, ks * 4 + lg);
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = gn_frag(As, wm + i * 16 + lr, ks * 4 + lg);
#endif
#ifdef GN_ABL_NOMFMA
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i].u4.x), "v"(fa[i].u4.w));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j].u4.x), "v"(fb[j].u4.w));
#else
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);   // swapped: D[n][m]
#endif
            }
#ifdef GN_TRACE
            {
                // the MFMAs have only been ISSUED: make the timestamp wait for the accumulators (a dependent no-op)
                asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[TM - 1][TN - 1][3]));
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();
                tr_wait += t1 - t0; tr_bar += t2 - t1; tr_issue += t3 - t2; tr_comp += t4 - t3; ++tr_stages;
            }
#endif
        }
        GN_T(te0);

        // ---- epilogue of tile q, from registers.  Swapped MFMA + permuted B rows: acc[i][j][r] =
        // C[m0 + wm + i*16 + lr][n0 + wn + (j>>1)*32 + lg*8 + (j&1)*4 + r]: the accumulator pair (2h, 2h+1) of a lane is
        // 8 consecutive columns = one 16-byte store, and the 4 lanes of a row write 64 contiguous bytes per instruction
        const int64_t m0 = (int64_t)tile_m(tq) * BM;
        const int n0 = tile_n(tq) * BN;
        const int nl = n0 + wn + lg * 8;                      // this lane's first column (of the pair h = 0)
        const bool edge = m0 + BM > M || n0 + BN > N;         // wave-uniform
#ifdef GN_ABL_NOSTORE
        if (m0 < 0)
#endif
#pragma unroll
        for (int h = 0; h < TN / 2; ++h) {
            const int nh = nl + h * 32;
            if (nh >= N) continue;                            // N % 8 == 0: groups of 8 columns are in or out together
            float bn[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bn[e] = 0.f;
            float sn[8];
            if constexpr (FP8) {
                const float4 s0 = *reinterpret_cast<const float4*>(sb + nh), s1 = *reinterpret_cast<const float4*>(sb + nh + 4);
                sn[0] = s0.x; sn[1] = s0.y; sn[2] = s0.z; sn[3] = s0.w; sn[4] = s1.x; sn[5] = s1.y; sn[6] = s1.z; sn[7] = s1.w;
            }
            if (HAS_BIAS) {
                if (bias_lds) {
                    const uint4 bv = *reinterpret_cast<const uint4*>(bias_s + nh);
                    const uint32_t bw[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) bn[e] = (e & 1) ? gn_hi(bw[e >> 1]) : gn_lo(bw[e >> 1]);
                } else {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias + nh), b1 = *reinterpret_cast<const float4*>(bias + nh + 4);
                    const float bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) bn[e] = bf2f(f2bf(bw[e]));
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int64_t m = m0 + wm + i * 16 + lr;
                if (m >= M) continue;
                if constexpr (EPI == GN_EPI_PARTIAL) {            // slice zq of the contraction: fp32 sums to slab zq
                    float* pz = partial + ((int64_t)zq * M + m) * N + nh;
                    *reinterpret_cast<f32x4_t*>(pz) = acc[i][2 * h];
                    *reinterpret_cast<f32x4_t*>(pz + 4) = acc[i][2 * h + 1];
                    continue;
                }
                const int64_t g = m * ldc + nh;
                float v[8];
                if constexpr (FP8) {
                    const float sm = sa[m];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(acc[i][2 * h + (e >> 2)][e & 3] * sm, sn[e], bn[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * h + (e >> 2)][e & 3] + bn[e];
                }
                if (EPI == GN_EPI_GELU) {
                    *reinterpret_cast<uint4*>(c2 + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                                                   pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2_t r = gelu_erf2((f32x2_t){v[e], v[e + 1]});
                        v[e] = r.x; v[e + 1] = r.y;
                    }
                }
                if (EPI == GN_EPI_GELUD) {                    // c2 = GELU'(pre): cdf and exp(-x^2/2) are in hand already
                    float d[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2_t x2 = {v[e], v[e + 1]};
                        f32x2_t cdf, ex;
                        gelu_parts2(x2, cdf, ex);
                        const f32x2_t r = x2 * cdf;
                        const f32x2_t dd = __builtin_elementwise_fma(x2 * (f32x2_t){0.3989422804014327f, 0.3989422804014327f}, ex, cdf);
                        v[e] = r.x; v[e + 1] = r.y;
                        d[e] = dd.x; d[e + 1] = dd.y;
                    }
                    *reinterpret_cast<uint4*>(c2 + g) = make_uint4(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]),
                                                                   pack2bf(d[4], d[5]), pack2bf(d[6], d[7]));
                }
                if (EPI == GN_EPI_MUL) {
                    const uint4 p = *reinterpret_cast<const uint4*>(aux + g);
                    const uint32_t pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) { v[e] *= gn_lo(pv[e >> 1]); v[e + 1] *= gn_hi(pv[e >> 1]); }
                }
                if (EPI == GN_EPI_DGELU) {
                    const uint4 p = *reinterpret_cast<const uint4*>(aux + g);
                    const uint32_t pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2_t r = gelu_erf_grad2((f32x2_t){gn_lo(pv[e >> 1]), gn_hi(pv[e >> 1])});
                        v[e] *= r.x; v[e + 1] *= r.y;
                    }
                }
                *reinterpret_cast<uint4*>(c + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                                              pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
            }
        }
#ifdef GN_ABL_NOSTORE
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j][0]), "v"(acc[i][j][3]));
        gn_wait_vm<0>();
#endif
        if (edge || nst < R - 1) {                            // fewer than S stores were issued (edge), or a second epilogue
            gn_wait_vm<0>();                                  // could fall into the window: drain, plain counts are valid again
            post_epi = 0;
        } else {
            post_epi = R - 1;
        }
#ifdef GN_TRACE
        tr_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
    }
#ifdef GN_TRACE
    if (gn_trace_buf && tid == 0) {
        unsigned long long* o = gn_trace_buf + (size_t)blockIdx.x * 16;
        o[0] = tr_wait; o[1] = tr_bar; o[2] = tr_issue; o[3] = tr_comp; o[4] = tr_epi; o[5] = tr_stages;
        o[6] = __builtin_amdgcn_s_memtime() - tr_start; o[7] = tr_start;
    }
#endif
}


// ---------------------------------------------------------------------------------------------------
// Wave-specialised version: NPROD producer waves issue ALL the LDS-DMA pieces, the WAVES_M x WAVES_N consumer waves only
// read fragments and multiply.  Why (tools/probes/gemm_lab.cpp, per-stage cycle sums of gemm_nt_kernel): a wave is blocked
// ~60 cycles per piece it issues — the CU's vector-memory path takes 1 KiB per ~16 cycles, i.e. 64 B/clk, and the 4-8 waves
// of a workgroup burst all the pieces of a stage at once — and it feeds no MFMA meanwhile; then the memory path idles while
// the waves multiply.  Issue and multiply each cost ~40 % of a stage, serially (64 x 128, 4 waves: 385 + 337 of 870 cycles;
// 128 x 128: 596 + 697), whatever the order (pieces interleaved with the MFMAs, or half the waves issuing late: both
// measured, both slower).  With the roles on different waves the matrix pipe and the memory path run side by side: a stage
// costs max(pieces x ~16-20 cycles, MFMAs x 16 cycles) instead of the sum.
// One barrier per stage for everybody: B_s says "stage s has landed and the slot of stage s-1 is free".  Producers wait
// (counted vmcnt) for their pieces of stage s before B_s and issue stage s+R-1 after it; consumers read slot s % R after it.
// Only the producers' vmcnt is ever waited on inside the pipeline, so the consumers' epilogue stores need no bookkeeping.
template <int BM, int BN, int WAVES_M, int WAVES_N, int NPROD, int R, int EPI>
__global__ void __launch_bounds__(64 * (WAVES_M * WAVES_N + NPROD), 1)
    gemm_ws_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ bias,
                   const bf16_t* __restrict__ aux, bf16_t* __restrict__ c, bf16_t* __restrict__ c2, int64_t M, int N, int K,
                   int64_t lda, int64_t ldb, int64_t ldc, int tilesN, int nmblk, int pc, int splitk, float* __restrict__ partial) {
    constexpr int NC = WAVES_M * WAVES_N, GN_THREADS = 64 * (NC + NPROD);
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 16, TN = WN / 16;
    constexpr int PA = BM / (8 * NPROD), PB = BN / (8 * NPROD), PP = PA + PB;     // pieces per producer wave and stage
    constexpr bool HAS_BIAS = EPI == GN_EPI_BIAS || EPI == GN_EPI_GELU || EPI == GN_EPI_GELUD;
    static_assert((R - 2) * PP <= 63, "vmcnt immediate");
    static_assert(TN % 2 == 0 && R >= 2 && BM % (8 * NPROD) == 0 && BN % (8 * NPROD) == 0, "tile shape");
    __shared__ __attribute__((aligned(1024))) unsigned char ring[R * STAGE + (HAS_BIAS ? GN_MAX_BIAS * 2 : 0)];
    bf16_t* bias_s = reinterpret_cast<bf16_t*>(ring + R * STAGE);

    GN_T(t_entry);
    const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, lr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
    const int pr = 8 / pc, xr = xcd / pc, xc = xcd - xr * pc;
#ifdef GN_TRACE
    asm volatile("" ::"s"(pr));                               // the first use of a kernel argument: its load has returned
    const unsigned long long t_args = __builtin_amdgcn_s_memtime();
#endif
    const int my_mblks = (nmblk - xr + pr - 1) / pr;
    const int my_tns = (tilesN - xc + pc - 1) / pc;
    const int my_units = my_mblks * my_tns * splitk;
    if (slot >= my_units || my_tns <= 0) return;
    auto tile_m = [&](int t) { return xr + pr * (t / my_tns); };
    auto tile_n = [&](int t) { return xc + pc * (t % my_tns); };
    const int nst_all = K / GN_BK;
    auto unit_span = [&](int u, int& kb, int& cnt) {
        const int t = u / splitk, z = u - t * splitk;
        kb = (nst_all * z) / splitk;
        cnt = (nst_all * (z + 1)) / splitk - kb;
        return t;
    };
    const unsigned ring_base = __builtin_amdgcn_readfirstlane(
        (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&ring[0]);

    if (wave >= NC) {
        // ------------------------------------------------------------------ producer
        const int pw = wave - NC;
        GnOper<BM, NPROD> A;
        GnOper<BN, NPROD> B;
        int qn = slot, issued = 0, icnt = 0;
        unsigned ibase = ring_base;
        auto init_unit = [&](int u) {
            int kb;
            const int t = unit_span(u, kb, icnt);
            gn_init<BM, NPROD, 0>(A, a + (int64_t)kb * GN_BK, (int64_t)tile_m(t) * BM, M, lda, pw, lane);
            gn_init<BN, NPROD, TN>(B, b + (int64_t)kb * GN_BK, tile_n(t) * BN, N, ldb, pw, lane);
        };
        GN_T(t_pre);
        init_unit(qn);
        auto issue_next = [&]() {
            if (qn >= my_units) return;
            const unsigned la = ibase + (unsigned)(pw * PA * 1024), lb = ibase + A_BYTES + (unsigned)(pw * PB * 1024);
#pragma unroll
            for (int j = 0; j < PA; ++j) gn_piece(la + j * 1024, A.voff[j], A.base);
#pragma unroll
            for (int j = 0; j < PB; ++j) gn_piece(lb + j * 1024, B.voff[j], B.base);
            A.base += GN_BK;
            B.base += GN_BK;
            ibase = ibase == ring_base + (R - 1) * STAGE ? ring_base : ibase + STAGE;
            if (++issued == icnt) {
                issued = 0;
                qn += nslot;
                if (qn < my_units) init_unit(qn);
            }
        };
        int total = 0;                                        // stages of all this workgroup's units
        for (int u = slot; u < my_units; u += nslot) {
            int kb, cnt;
            unit_span(u, kb, cnt);
            total += cnt;
        }
        GN_T(t_init);
#pragma unroll
        for (int d = 0; d < R - 1; ++d) issue_next();
        GN_T(t_primed);
#ifdef GN_TRACE
        unsigned long long tr_wait = 0, tr_issue = 0, t_first = 0;
#endif
        for (int s = 0; s < total; ++s) {
            // stage s has landed when at most the R-2 younger stages are outstanding; near the end fewer are in flight
            GN_T(t0);
            if (total - 1 - s >= R - 2) gn_wait_vm<(R - 2) * PP>();
            else gn_wait_vm<0>();
            GN_T(t1);
            __builtin_amdgcn_s_barrier();
            GN_T(t2);
            issue_next();
#ifdef GN_TRACE
            if (s == 0) t_first = t1;
            tr_wait += t1 - t0;
            tr_issue += __builtin_amdgcn_s_memtime() - t2;
#endif
        }
#ifdef GN_TRACE
        if (gn_trace_buf && pw == 0 && lane == 0) {
            unsigned long long* o = gn_trace_buf + (size_t)blockIdx.x * 16;
            o[0] = tr_wait; o[2] = tr_issue;
            o[8] = t_init - t_entry; o[9] = t_primed - t_init; o[10] = t_first - t_primed;
            o[12] = t_args - t_entry; o[13] = t_pre - t_args; o[14] = t_init - t_pre;
        }
#endif
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = (wave / WAVES_N) * WM, wn = (wave % WAVES_N) * WN;
    if (HAS_BIAS) {
        for (int i = tid * 4; i < N; i += 64 * NC * 4) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + i);
            *reinterpret_cast<uint2*>(bias_s + i) = make_uint2(pack2bf(bv.x, bv.y), pack2bf(bv.z, bv.w));
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): published by the first stage's barrier
    }
    int cslot = 0;
#ifdef GN_TRACE
    unsigned long long tr_bar = 0, tr_comp = 0, tr_epi = 0, tr_stages = 0;
    const unsigned long long tr_start = __builtin_amdgcn_s_memtime();
#endif
    int total = 0;                                            // stages of all this workgroup's units (as the producers count)
    for (int u = slot; u < my_units; u += nslot) {
        int kb, cnt;
        unit_span(u, kb, cnt);
        total += cnt;
    }
    // The fragments of a stage (both k halves: (TM + TN) x 2 x 16 bytes per lane) are held in REGISTERS, double-buffered:
    // while the MFMAs of stage s run on one set, the ds_reads of stage s+1 fill the other — a consumer wave is alone on its
    // SIMD, so nothing else would cover the ~120 cycles between a ds_read and its first use (measured: 755 cycles per
    // 512-cycle stage without this).  A wave arrives at barrier B_{s+1} once its reads of stage s have RETURNED, which is
    // what frees slot s for the producers' next stage.
    Frag8 fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];
    auto load_frags = [&](Frag8 (&fa)[2][TM], Frag8 (&fb)[2][TN]) {
        const unsigned char* As = ring + cslot * STAGE;
        const unsigned char* Bs = As + A_BYTES;
        cslot = cslot == R - 1 ? 0 : cslot + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[ks][j] = gn_frag(Bs, wn + j * 16 + lr, ks * 4 + lg);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[ks][i] = gn_frag(As, wm + i * 16 + lr, ks * 4 + lg);
        }
    };
    f32x4_t acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    };
    int q = slot, st = 0, gs = 0, nst, kb_q;
    int tq = unit_span(q, kb_q, nst);
    zero_acc();
    auto epilogue = [&]() {
        const int zq = q - tq * splitk;
        GN_T(te0);
        // ---- epilogue from registers (layout as in gemm_nt_kernel)
        const int64_t m0 = (int64_t)tile_m(tq) * BM;
        const int n0 = tile_n(tq) * BN;
        const int nl = n0 + wn + lg * 8;
#pragma unroll
        for (int h = 0; h < TN / 2; ++h) {
            const int nh = nl + h * 32;
            if (nh >= N) continue;
            float bn[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bn[e] = 0.f;
            if (HAS_BIAS) {
                const uint4 bv = *reinterpret_cast<const uint4*>(bias_s + nh);
                const uint32_t bw[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) bn[e] = (e & 1) ? gn_hi(bw[e >> 1]) : gn_lo(bw[e >> 1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int64_t m = m0 + wm + i * 16 + lr;
                if (m >= M) continue;
                if constexpr (EPI == GN_EPI_PARTIAL) {
                    float* pz = partial + ((int64_t)zq * M + m) * N + nh;
                    *reinterpret_cast<f32x4_t*>(pz) = acc[i][2 * h];
                    *reinterpret_cast<f32x4_t*>(pz + 4) = acc[i][2 * h + 1];
                    continue;
                }
                const int64_t g = m * ldc + nh;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * h + (e >> 2)][e & 3] + bn[e];
                if (EPI == GN_EPI_GELU) {
                    *reinterpret_cast<uint4*>(c2 + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                                                   pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2_t r = gelu_erf2((f32x2_t){v[e], v[e + 1]});
                        v[e] = r.x; v[e + 1] = r.y;
                    }
                }
                if (EPI == GN_EPI_GELUD) {
                    float d[8];
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const f32x2_t x2 = {v[e], v[e + 1]};
                        f32x2_t cdf, ex;
                        gelu_parts2(x2, cdf, ex);
                        const f32x2_t r = x2 * cdf;
                        const f32x2_t dd = __builtin_elementwise_fma(x2 * (f32x2_t){0.3989422804014327f, 0.3989422804014327f}, ex, cdf);
                        v[e] = r.x; v[e + 1] = r.y;
                        d[e] = dd.x; d[e + 1] = dd.y;
                    }
                    *reinterpret_cast<uint4*>(c2 + g) = make_uint4(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]),
                                                                   pack2bf(d[4], d[5]), pack2bf(d[6], d[7]));
                }
                if (EPI == GN_EPI_MUL || EPI == GN_EPI_DGELU) {
                    const uint4 pa = *reinterpret_cast<const uint4*>(aux + g);
                    const uint32_t pv[4] = {pa.x, pa.y, pa.z, pa.w};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        if (EPI == GN_EPI_MUL) { v[e] *= gn_lo(pv[e >> 1]); v[e + 1] *= gn_hi(pv[e >> 1]); }
                        else {
                            const f32x2_t r = gelu_erf_grad2((f32x2_t){gn_lo(pv[e >> 1]), gn_hi(pv[e >> 1])});
                            v[e] *= r.x; v[e + 1] *= r.y;
                        }
                    }
                }
                *reinterpret_cast<uint4*>(c + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                                              pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
            }
        }
#ifdef GN_TRACE
        tr_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
    };
    auto step = [&](Frag8 (&ca)[2][TM], Frag8 (&cb)[2][TN], Frag8 (&na)[2][TM], Frag8 (&nb)[2][TN]) {
        GN_T(t0);
        __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0): this stage's fragments are in registers
        if (gs + 1 < total) {
            __builtin_amdgcn_s_barrier();                     // B_{gs+1}: the next stage has landed, this one's slot is free
            load_frags(na, nb);
        }
        GN_T(t1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(cb[ks][j], ca[ks][i], acc[i][j]);   // swapped: D[n][m]
#ifdef GN_TRACE
        asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[TM - 1][TN - 1][3]));
        tr_bar += t1 - t0; tr_comp += __builtin_amdgcn_s_memtime() - t1; ++tr_stages;
#endif
        ++gs;
        if (++st == nst) {
            epilogue();
            q += nslot;
            st = 0;
            if (q < my_units) {
                tq = unit_span(q, kb_q, nst);
                zero_acc();
            }
        }
        return gs == total;
    };
    __builtin_amdgcn_s_barrier();                             // B_0
#ifdef GN_TRACE
    const unsigned long long t_b0 = __builtin_amdgcn_s_memtime();
#endif
    load_frags(fa0, fb0);
    while (true) {
        if (step(fa0, fb0, fa1, fb1)) break;
        if (step(fa1, fb1, fa0, fb0)) break;
    }
#ifdef GN_TRACE
    if (gn_trace_buf && tid == 0) {
        unsigned long long* o = gn_trace_buf + (size_t)blockIdx.x * 16;
        o[1] = tr_bar; o[3] = tr_comp; o[4] = tr_epi; o[5] = tr_stages;
        o[6] = __builtin_amdgcn_s_memtime() - t_entry; o[7] = tr_start; o[11] = t_b0 - t_entry;
    }
#endif
}

// Second kernel of a split-K launch: c = epilogue(sum over the S slices of partial[s][m][n]).  One thread per 8 consecutive
// columns (32-byte fp32 reads per slab, one 16-byte bf16 store); the slabs were written a kernel boundary ago, mostly still
// in L2 / the Infinity Cache.
template <int EPI>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ partial, int S, const float* __restrict__ bias,
                                                            const bf16_t* __restrict__ aux, bf16_t* __restrict__ c,
                                                            bf16_t* __restrict__ c2, int64_t M, int N, int64_t ldc) {
    const int n8 = N >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * n8) return;
    const int64_t m = idx / n8;
    const int nh = (int)(idx - m * n8) * 8;
    const float* p = partial + m * N + nh;
    const int64_t slab = M * (int64_t)N;
    f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(p), s1 = *reinterpret_cast<const f32x4_t*>(p + 4);
    for (int z = 1; z < S; ++z) {
        s0 += *reinterpret_cast<const f32x4_t*>(p + z * slab);
        s1 += *reinterpret_cast<const f32x4_t*>(p + z * slab + 4);
    }
    float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    constexpr bool HAS_BIAS = EPI == GN_EPI_BIAS || EPI == GN_EPI_GELU || EPI == GN_EPI_GELUD;
    if (HAS_BIAS) {                                           // the bias as the bf16 operand the one-pass kernel uses
        const float4 b0 = *reinterpret_cast<const float4*>(bias + nh), b1 = *reinterpret_cast<const float4*>(bias + nh + 4);
        const float bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf2f(f2bf(bw[e]));
    }
    const int64_t g = m * ldc + nh;
    if (EPI == GN_EPI_GELU) {
        *reinterpret_cast<uint4*>(c2 + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_t r = gelu_erf2((f32x2_t){v[e], v[e + 1]});
            v[e] = r.x; v[e + 1] = r.y;
        }
    }
    if (EPI == GN_EPI_GELUD) {
        float d[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const f32x2_t x2 = {v[e], v[e + 1]};
            f32x2_t cdf, ex;
            gelu_parts2(x2, cdf, ex);
            const f32x2_t r = x2 * cdf;
            const f32x2_t dd = __builtin_elementwise_fma(x2 * (f32x2_t){0.3989422804014327f, 0.3989422804014327f}, ex, cdf);
            v[e] = r.x; v[e + 1] = r.y;
            d[e] = dd.x; d[e + 1] = dd.y;
        }
        *reinterpret_cast<uint4*>(c2 + g) = make_uint4(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]), pack2bf(d[4], d[5]), pack2bf(d[6], d[7]));
    }
    if (EPI == GN_EPI_MUL || EPI == GN_EPI_DGELU) {
        const uint4 pa = *reinterpret_cast<const uint4*>(aux + g);
        const uint32_t pv[4] = {pa.x, pa.y, pa.z, pa.w};
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            if (EPI == GN_EPI_MUL) { v[e] *= gn_lo(pv[e >> 1]); v[e + 1] *= gn_hi(pv[e >> 1]); }
            else {
                const f32x2_t r = gelu_erf_grad2((f32x2_t){gn_lo(pv[e >> 1]), gn_hi(pv[e >> 1])});
                v[e] *= r.x; v[e + 1] *= r.y;
            }
        }
    }
    *reinterpret_cast<uint4*>(c + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

int gn_launch_reduce(int epi, hipStream_t st, const float* partial, int S, const float* bias, const bf16_t* aux, bf16_t* c, bf16_t* c2,
                     int64_t M, int N, int64_t ldc) {
    const int64_t groups = M * (N >> 3);
    const unsigned grid = (unsigned)((groups + 255) / 256);
#define GN_GO(E) hipLaunchKernelGGL((splitk_reduce_kernel<E>), dim3(grid), dim3(256), 0, st, partial, S, bias, aux, c, c2, M, N, ldc)
    switch (epi) {
        case GN_EPI_NONE: GN_GO(GN_EPI_NONE); break;
        case GN_EPI_BIAS: GN_GO(GN_EPI_BIAS); break;
        case GN_EPI_GELU: GN_GO(GN_EPI_GELU); break;
        case GN_EPI_DGELU: GN_GO(GN_EPI_DGELU); break;
        case GN_EPI_GELUD: GN_GO(GN_EPI_GELUD); break;
        case GN_EPI_MUL: GN_GO(GN_EPI_MUL); break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef GN_GO
    return clv_check_launch();
}

// ---------------------------------------------------------------------------------------------------
// One-tile-per-workgroup version (probe: CLV_GEMM_TILE=lean): ONE 128 x BN tile per 256-thread workgroup (2 x 2 waves), 64-deep stages in a double
// buffer, two workgroups per CU.  What the SQ counters showed about the first loops applies here as it did to the
// weight-gradient kernel: per-piece pointer bookkeeping cost several times the 32 MFMAs of a stage.  So: the loop is
// unrolled over the two slots (every LDS address = per-lane constant + immediate), the 8 DMA pieces of a stage are one
// asm block in the SGPR-base + 32-bit-VGPR-offset form (a stage advance = two scalar adds per operand), rows past the
// edge are clamped.  Dispatch order keeps the N tiles of a row block on one XCD (as above).
template <int BN, int EPI>
__global__ void __launch_bounds__(256, 2) gemm_nt_lean_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b,
                                                              const float* __restrict__ bias, const bf16_t* __restrict__ aux,
                                                              bf16_t* __restrict__ c, bf16_t* __restrict__ c2, int64_t M, int N,
                                                              int K, int64_t lda, int64_t ldb, int64_t ldc, int tilesN,
                                                              int nmblk) {
    constexpr int BM = 128, WAVES = 4;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    constexpr int PA = BM / (8 * WAVES), PB = BN / (8 * WAVES);            // DMA pieces per wave: 4 and 4 (2)
    constexpr bool HAS_BIAS = EPI == GN_EPI_BIAS || EPI == GN_EPI_GELU;
    __shared__ __attribute__((aligned(1024))) unsigned char ring[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
    const int mblk = xcd + 8 * (xslot / tilesN), tn = xslot % tilesN;
    if (mblk >= nmblk) return;
    const int64_t m0 = (int64_t)mblk * BM;
    const int n0 = tn * BN;
    const int wm = (wave >> 1) * WM, wn = (wave & 1) * WN;

    // per-lane source offsets in bytes from the tile's first row (fixed for the whole K loop); the wave's pieces are
    // consecutive 1-KiB blocks of the operand's LDS image
    unsigned va[PA], vb[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int r = (wave * PA + j) * 8 + (lane >> 3);                   // LDS row = global row (A)
        const int cch = (lane & 7) ^ (r & 7);
        int64_t gr = m0 + r;
        gr = gr < M ? gr : M - 1;
        va[j] = (unsigned)(((gr - m0) * lda + cch * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int r = (wave * PB + j) * 8 + (lane >> 3);                   // LDS row; global row permuted (see gn_init)
        const int cch = (lane & 7) ^ (r & 7);
        const int hh = r / WN, p = r % WN, jj = p >> 4, x = p & 15;
        int g = n0 + hh * WN + (jj >> 1) * 32 + (x >> 2) * 8 + (jj & 1) * 4 + (x & 3);
        g = g < N ? g : N - 1;
        vb[j] = (unsigned)(((int64_t)(g - n0) * ldb + cch * 8) * 2);
    }
    const bf16_t* abase = a + m0 * lda;                                    // wave-uniform (SGPR pairs), + 64 per stage
    const bf16_t* bbase = b + (int64_t)n0 * ldb;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&ring[0];
    const unsigned lds_a = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wave * PA * 1024));
    const unsigned lds_b = __builtin_amdgcn_readfirstlane(lds0 + A_BYTES + (unsigned)(wave * PB * 1024));

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // per-lane fragment addresses inside an operand image: row (w? + t*16 + lr), k chunk ks*4 + lg, XOR (row & 7);
    // (t*16 + lr) & 7 == lr & 7 and wm, wn are multiples of 8, so one base per k half + immediates t * 2048
    int fa0[2], fb0[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        fa0[ks] = (wm + lr) * 128 + (((ks * 4 + lg) ^ (lr & 7)) << 4);
        fb0[ks] = A_BYTES + (wn + lr) * 128 + (((ks * 4 + lg) ^ (lr & 7)) << 4);
    }
    auto compute = [&](const unsigned char* st) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            Frag8 fa[TM], fb[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j].u4 = *reinterpret_cast<const uint4*>(st + fb0[ks] + j * 2048);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i].u4 = *reinterpret_cast<const uint4*>(st + fa0[ks] + i * 2048);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(fb[j], fa[i], acc[i][j]);   // swapped: D[n][m]
        }
    };
#define GNL_ISSUE(SLOT)                                  \
    gn_dma_block<PA>(lds_a + (SLOT) * STAGE, va, abase); \
    gn_dma_block<PB>(lds_b + (SLOT) * STAGE, vb, bbase); \
    abase += GN_BK;                                      \
    bbase += GN_BK;
    const int nst = K / GN_BK;
    GNL_ISSUE(0)
    int st = 0;
    for (; st + 2 <= nst; st += 2) {
        gn_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                     // stage st landed everywhere; slot 1 is drained
        GNL_ISSUE(1)
        compute(ring);
        gn_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        if (st + 2 < nst) { GNL_ISSUE(0) }
        compute(ring + STAGE);
    }
    if (st < nst) {                                       // odd stage count: the last one sits in slot 0
        gn_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        compute(ring);
    }
#undef GNL_ISSUE

    // ---- epilogue from registers (see gemm_nt_kernel): accumulator pair (2h, 2h+1) = 8 consecutive columns
    const int nl = n0 + wn + lg * 8;
#pragma unroll
    for (int h = 0; h < TN / 2; ++h) {
        const int nh = nl + h * 32;
        if (nh >= N) continue;
        float bn[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bn[e] = 0.f;
        if (HAS_BIAS) {                                   // bias as the bf16 operand the library GEMM took
            const float4 b0 = *reinterpret_cast<const float4*>(bias + nh), b1 = *reinterpret_cast<const float4*>(bias + nh + 4);
            const float bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) bn[e] = bf2f(f2bf(bw[e]));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int64_t m = m0 + wm + i * 16 + lr;
            if (m >= M) continue;
            const int64_t g = m * ldc + nh;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[i][2 * h + (e >> 2)][e & 3] + bn[e];
            if (EPI == GN_EPI_GELU) {
                *reinterpret_cast<uint4*>(c2 + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                                               pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2_t r = gelu_erf2((f32x2_t){v[e], v[e + 1]});
                    v[e] = r.x; v[e + 1] = r.y;
                }
            }
            if (EPI == GN_EPI_DGELU) {
                const uint4 p = *reinterpret_cast<const uint4*>(aux + g);
                const uint32_t pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2_t r = gelu_erf_grad2((f32x2_t){gn_lo(pv[e >> 1]), gn_hi(pv[e >> 1])});
                    v[e] *= r.x; v[e + 1] *= r.y;
                }
            }
            *reinterpret_cast<uint4*>(c + g) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]),
                                                          pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        }
    }
}

template <int BN>
int gn_launch_lean(int epi, hipStream_t st, unsigned grid, const bf16_t* a, const bf16_t* b, const float* bias,
                   const bf16_t* aux, bf16_t* c, bf16_t* c2, int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                   int tilesN, int nmblk) {
#define GN_GO(E)                                                                                                       \
    hipLaunchKernelGGL((gemm_nt_lean_kernel<BN, E>), dim3(grid), dim3(256), 0, st, a, b, bias, aux, c, c2, M, N, K, lda, \
                       ldb, ldc, tilesN, nmblk)
    switch (epi) {
        case GN_EPI_NONE: GN_GO(GN_EPI_NONE); break;
        case GN_EPI_BIAS: GN_GO(GN_EPI_BIAS); break;
        case GN_EPI_GELU: GN_GO(GN_EPI_GELU); break;
        case GN_EPI_DGELU: GN_GO(GN_EPI_DGELU); break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef GN_GO
    return clv_check_launch();
}

// Batched 2-D transposes of bf16 matrices (the W^T shadows): one 64 x 64 tile per workgroup through LDS, 16-byte
// accesses on both sides.  Entry e: src [rows][cols] -> dst [cols][rows]; tile_begin = prefix sum of tile counts.
// The tile is stored with its 8-element column groups XOR-swizzled by the row group (element (r, c) at column c ^ (r & 56)):
// the column-wise 2-byte reads of the second half (8 row groups x 8 neighbouring columns per wave) then fall on 32 distinct
// banks instead of 2 (rows 8 apart are 32 banks apart at any 16-byte-aligned row pitch).
struct TrEntry {
    int64_t src_off, dst_off;                                 // element offsets into src_base / dst_base
    int32_t rows, cols, tile_begin, tiles_c;
};

__global__ void __launch_bounds__(256) transpose_batch_kernel(const bf16_t* __restrict__ src_base, bf16_t* __restrict__ dst_base,
                                                              const TrEntry* __restrict__ table, int n_entries) {
    __shared__ bf16_t tile[64][64 + 8];
    const int t = blockIdx.x;
    int lo = 0, hi = n_entries - 1;                           // the entry whose tile range holds t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].tile_begin <= t) lo = mid; else hi = mid - 1;
    }
    const TrEntry e = table[lo];
    const int lt = t - e.tile_begin, tr = lt / e.tiles_c, tc = lt % e.tiles_c;
    const int r0 = tr * 64, c0 = tc * 64;
    const bf16_t* src = src_base + e.src_off;
    bf16_t* dst = dst_base + e.dst_off;
    const bool vec = (e.rows % 8 == 0) && (e.cols % 8 == 0);
    for (int idx = threadIdx.x; idx < 64 * 8; idx += 256) {
        const int r = idx >> 3, c8 = (idx & 7) * 8;
        if (vec) {
            if (r0 + r < e.rows && c0 + c8 < e.cols)
                *reinterpret_cast<uint4*>(&tile[r][c8 ^ (r & 56)]) =
                    *reinterpret_cast<const uint4*>(src + (int64_t)(r0 + r) * e.cols + c0 + c8);
        } else {
            for (int q = 0; q < 8; ++q)
                if (r0 + r < e.rows && c0 + c8 + q < e.cols)
                    tile[r][(c8 + q) ^ (r & 56)] = src[(int64_t)(r0 + r) * e.cols + c0 + c8 + q];
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * 8; idx += 256) {
        const int cc = idx >> 3, r8 = (idx & 7) * 8;          // output row = source column
        if (c0 + cc >= e.cols) continue;
        if (vec) {
            if (r0 + r8 < e.rows) {
                u16x8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) o.v[q] = tile[r8 + q][cc ^ r8];
                *reinterpret_cast<u16x8*>(dst + (int64_t)(c0 + cc) * e.rows + r0 + r8) = o;
            }
        } else {
            for (int q = 0; q < 8; ++q)
                if (r0 + r8 + q < e.rows) dst[(int64_t)(c0 + cc) * e.rows + r0 + r8 + q] = tile[r8 + q][cc ^ r8];
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int R>
int gn_launch(int epi, hipStream_t st, unsigned grid, const bf16_t* a, const bf16_t* b, const float* bias, const bf16_t* aux,
              bf16_t* c, bf16_t* c2, int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int tilesN, int nmblk,
              int pc = 1, int splitk = 1, int rot = 0, float* partial = nullptr) {
#define GN_GO(E)                                                                                                         \
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, R, E>), dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, st, \
                       a, b, bias, aux, c, c2, M, N, K, lda, ldb, ldc, tilesN, nmblk, (const float*)nullptr,             \
                       (const float*)nullptr, pc, splitk, rot, partial)
    if (splitk > 1) {                                         // K slices leave fp32 partials; the epilogue runs in the reduce
        GN_GO(GN_EPI_PARTIAL);
        const int rc = clv_check_launch();
        return rc != CLV_OK ? rc : gn_launch_reduce(epi, st, partial, splitk, bias, aux, c, c2, M, N, ldc);
    }
    switch (epi) {
        case GN_EPI_NONE: GN_GO(GN_EPI_NONE); break;
        case GN_EPI_BIAS: GN_GO(GN_EPI_BIAS); break;
        case GN_EPI_GELU: GN_GO(GN_EPI_GELU); break;
        case GN_EPI_DGELU: GN_GO(GN_EPI_DGELU); break;
        case GN_EPI_GELUD: GN_GO(GN_EPI_GELUD); break;
        case GN_EPI_MUL: GN_GO(GN_EPI_MUL); break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef GN_GO
    return clv_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int R>
int gn_launch_fp8(int epi, hipStream_t st, unsigned grid, const bf16_t* a, const bf16_t* b, const float* bias, bf16_t* c,
                  bf16_t* c2, int64_t M, int N, int K2, int64_t lda2, int64_t ldb2, int64_t ldc, int tilesN, int nmblk,
                  const float* sa, const float* sb, int pc = 1) {
#define GN_GO(E)                                                                                                         \
    hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WAVES_M, WAVES_N, R, E, true>), dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, \
                       st, a, b, bias, (const bf16_t*)nullptr, c, c2, M, N, K2, lda2, ldb2, ldc, tilesN, nmblk, sa, sb, pc)
    switch (epi) {
        case GN_EPI_NONE: GN_GO(GN_EPI_NONE); break;
        case GN_EPI_BIAS: GN_GO(GN_EPI_BIAS); break;
        case GN_EPI_GELUD: GN_GO(GN_EPI_GELUD); break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef GN_GO
    return clv_check_launch();
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int NPROD, int R>
int gn_launch_ws(int epi, hipStream_t st, unsigned grid, const bf16_t* a, const bf16_t* b, const float* bias, const bf16_t* aux,
                 bf16_t* c, bf16_t* c2, int64_t M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc, int tilesN, int nmblk,
                 int pc, int splitk, float* partial) {
#define GN_GO(E)                                                                                                          \
    hipLaunchKernelGGL((gemm_ws_kernel<BM, BN, WAVES_M, WAVES_N, NPROD, R, E>), dim3(grid),                              \
                       dim3(64 * (WAVES_M * WAVES_N + NPROD)), 0, st, a, b, bias, aux, c, c2, M, N, K, lda, ldb, ldc, tilesN, \
                       nmblk, pc, splitk, partial)
    if (splitk > 1) {
        GN_GO(GN_EPI_PARTIAL);
        const int rc = clv_check_launch();
        return rc != CLV_OK ? rc : gn_launch_reduce(epi, st, partial, splitk, bias, aux, c, c2, M, N, ldc);
    }
    switch (epi) {
        case GN_EPI_NONE: GN_GO(GN_EPI_NONE); break;
        case GN_EPI_BIAS: GN_GO(GN_EPI_BIAS); break;
        case GN_EPI_GELU: GN_GO(GN_EPI_GELU); break;
        case GN_EPI_DGELU: GN_GO(GN_EPI_DGELU); break;
        case GN_EPI_GELUD: GN_GO(GN_EPI_GELUD); break;
        case GN_EPI_MUL: GN_GO(GN_EPI_MUL); break;
        default: return CLV_ERR_UNSUPPORTED;
    }
#undef GN_GO
    return clv_check_launch();
}

// Column partitions of the XCD split (see the kernel): the pc in {1, 2, 4, 8} with the smallest per-XCD operand footprint
// A_bytes * pc / 8 + B_bytes / pc, among those that leave every XCD at least one column tile.
int gn_pick_pc(int64_t a_bytes, int64_t b_bytes, int tilesN) {
    static const int forced = getenv("CLV_GEMM_PC") ? atoi(getenv("CLV_GEMM_PC")) : 0;
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced <= tilesN ? forced : 1;
    int best = 1;
    int64_t best_ws = a_bytes / 8 + b_bytes;
    for (int pc = 2; pc <= 8; pc *= 2) {
        if (pc > tilesN) break;
        const int64_t ws = a_bytes * pc / 8 + b_bytes / pc;
        if (ws < best_ws) { best = pc; best_ws = ws; }
    }
    return best;
}

}  // namespace

extern "C" int clv_gemm_nt_fp8_supported(int64_t M, int32_t N, int32_t K) {
    return M >= 1 && N >= 64 && N % 8 == 0 && N <= GN_MAX_BIAS && K >= 128 && K % 128 == 0;
}

extern "C" int clv_gemm_nt_fp8(const void* a8, const void* b8, const float* sa, const float* sb, const float* bias, void* c,
                               void* c2, int64_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                               int32_t epilogue, void* stream) {
    if (!a8 || !b8 || !sa || !sb || !c || M <= 0 || N <= 0 || K <= 0) return CLV_ERR_ARG;
    if (!clv_gemm_nt_fp8_supported(M, N, K)) return CLV_ERR_UNSUPPORTED;
    if ((lda & 15) || (ldb & 15) || (ldc & 7) || lda < K || ldb < K || ldc < N) return CLV_ERR_ARG;
    if ((((uintptr_t)a8) | ((uintptr_t)b8) | ((uintptr_t)c) | ((uintptr_t)c2) | ((uintptr_t)bias) | ((uintptr_t)sb)) & 15)
        return CLV_ERR_ARG;
    if ((epilogue == GN_EPI_BIAS || epilogue == GN_EPI_GELUD) && !bias) return CLV_ERR_ARG;
    if (epilogue == GN_EPI_GELUD && !c2) return CLV_ERR_ARG;
    int BM = 128;
    if (K >= 1024 && ((M + 127) / 128) * ((N + 127) / 128) <= 384) BM = 64;      // same rule as clv_gemm_nt (K in bytes here)
    const int BN = 128;
    const int tilesN = (N + BN - 1) / BN;
    const int nmblk = (int)((M + BM - 1) / BM);
    const int pc = gn_pick_pc((int64_t)M * K, (int64_t)N * K, tilesN);
    const int max_tiles_xcd = ((nmblk + 8 / pc - 1) / (8 / pc)) * ((tilesN + pc - 1) / pc);
    const unsigned grid = (unsigned)(8 * (max_tiles_xcd < 64 ? max_tiles_xcd : 64));
    hipStream_t st = (hipStream_t)stream;
    // the staging code moves bytes: hand it the operands in 2-byte units
#define GN_ARGS8 epilogue, st, grid, (const bf16_t*)a8, (const bf16_t*)b8, bias, (bf16_t*)c, (bf16_t*)c2, M, N, K / 2, lda / 2, \
                 ldb / 2, ldc, tilesN, nmblk, sa, sb, pc
    static const int w8 = getenv("CLV_GEMM_W8") ? atoi(getenv("CLV_GEMM_W8")) : 1;     // eight waves, as the bf16 class
    if (BM == 128 && w8) return gn_launch_fp8<128, 128, 2, 4, 2>(GN_ARGS8);
    if (BM == 128) return gn_launch_fp8<128, 128, 2, 2, 2>(GN_ARGS8);
    return gn_launch_fp8<64, 128, 2, 2, 3>(GN_ARGS8);
#undef GN_ARGS8
}

extern "C" int clv_gemm_nt_supported(int64_t M, int32_t N, int32_t K) {
    return M >= 1 && N >= 64 && N % 8 == 0 && N <= (1 << 20) && K >= 64 && K % GN_BK == 0;
}

namespace {

// Tile class and K split of one clv_gemm_nt call.
struct GnPlan {
    int BM, BN, W;
    bool r2;
    int splitk, rot;
    int ws;                 // 0: gemm_nt_kernel; 1: gemm_ws_kernel<128,128, 8 consumers + 2 producers>; 2: <64,128, 4 + 2>
};

int gn_env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Measured with cold weights (tools/probes/gemm_lab.cpp, us per launch):
//  * few rows (M <= 1024: the text tower) — the wave-specialised 64 x 128 class (fc1 10.2 vs 11.2), and for a long
//    contraction (K >= 1536: 48 tiles x 36-48 stages on 256 CUs) FOUR K slices as work units of their own + the reduce kernel
//    (fc2 15.6 vs 25.9, qkv input gradient 14.3 vs 20.5).  The slices cost 4 fp32 slabs of M x N written and read once, which
//    is why rows in the thousands never split: at M = 3 136 three slices of fc2 take 40 us against 29.
//  * <= 256 tiles of 128 x 128 and K >= 1536 (fc2 / fc1-dgrad / qkv-dgrad / PatchMerging of Swin stage 3, fusion encoder):
//    one workgroup per CU anyway, so the wave-specialised 128 x 128 class with its ring of 4 (fc2 s3 29.0 vs 33.9, fusion
//    fc2 29.3 vs 34.9, dqkv s3 23.0 vs 27.2, merge s3 17.9 vs 20.3).
//  * otherwise the one-kernel classes: 64 x 128 (<= 384 tiles, K >= 512) or 128 x 128 on eight waves, two workgroups per CU
//    (the wave-specialised classes tie or lose there: 20.8 vs 19.7 for qkv s3, 34.5 vs 32.1 for fc2 s1).
GnPlan gn_plan(int64_t M, int N, int K, bool allow_split) {
    GnPlan p{128, 128, 4, false, 1, 0, 0};
    const int64_t tiles128 = ((M + 127) / 128) * ((N + 127) / 128);
    const int64_t tiles64 = ((M + 63) / 64) * ((N + 127) / 128);
    if (K >= 512 && tiles128 <= 384) p.BM = 64;
    // probe knobs, read per call (a few getenv: ~0.2 us of host time; the step replays hipGraphs)
    const int w8 = gn_env_int("CLV_GEMM_W8", 1);
    const int force_s = gn_env_int("CLV_GEMM_SPLITK", 0);                 // 0: auto, 1: off, n: n slices where allowed
    const int ws = gn_env_int("CLV_GEMM_WS", 3);                          // bit 0: the 128 x 128 class, bit 1: the few-row class
    const int nst = K / GN_BK;
    p.rot = gn_env_int("CLV_GEMM_ROT", 0) && nst >= 3;                    // measured neutral (hot and cold operands): off
    if ((ws & 2) && M <= 1024 && K >= 512 && tiles64 <= 256 && N <= GN_MAX_BIAS) {
        p.ws = 2;
        p.BM = 64;
        if (allow_split && force_s != 1 && K >= 1536 && tiles64 <= 96) {
            // enough slices to put ~256 units on the chip, each >= 12 stages deep (measured: 4 slices for the text tower's
            // 48-tile K = 3 072 layers; the MLM decoder's input gradient contracts over 30 528 vocabulary entries)
            int sk = (int)((256 + tiles64 / 2) / tiles64);
            sk = sk < 4 ? 4 : (sk > 16 ? 16 : sk);
            // few tiles, very long contraction (the decoder: 24 tiles, 477 stages): past 192 units the launch takes a second
            // round on part of the chip — 6 or 8 slices 32 us, 9-11 slices 44-46 us
            if (tiles64 <= 32)
                while (sk > 4 && tiles64 * sk > 192) --sk;
            while (sk > 2 && nst / sk < 12) --sk;
            p.splitk = sk;
        }
    } else if ((ws & 1) && tiles128 <= 256 && K >= 1536 && N <= GN_MAX_BIAS) {
        p.ws = 1;
        p.BM = 128;
    }
    // Outputs of 192 / 384 / 576 columns (every width of Swin-T's stages 1-2) on many rows: 128 x 192 tiles — 128-column
    // tiles leave a quarter of the MFMAs of N = 192 on clamped columns, and 98 x 3 tiles of N = 384 run as two rounds on 256
    // CUs where 98 x 2 run as one (cold weights, us: fc2 s1 30.3 -> 26.1, dqkv s1 25.4 -> 21.5, fc2 s2 28.9 -> 23.1, dqkv s2
    // 22.9 -> 18.8, merge s2 17.5 -> 14.9, proj s1 / s2 14.1 -> 12.7 / 11.1 -> 10.3; library 23.1 / 20.6 / 21.8 / 17.8 / 13.0 /
    // 13.4 / 12.8).  N = 1 152 and the M ~ 3 000 shapes lose on it (qkv s2 20.6 -> 23.9, fc2 s3 28.0 -> 39.0).
    // Where the 128 x 192 tiles do not fill the chip once (12 544 rows x 384 columns: 196 tiles), 64 x 192 tiles on four waves,
    // two workgroups per CU: fc2 s2 23.0 -> 21.9, dqkv s2 18.6 -> 17.7, merge s2 14.7 -> 13.5, proj s2 10.2 -> 9.6.
    if (!p.ws && gn_env_int("CLV_GEMM_T192", 1) && N % 192 == 0 && N <= 576 && M >= 8192) {
        const bool small = ((M + 127) / 128) * (N / 192) <= 256;
        p.BM = small ? 64 : 128;
        p.BN = 192;
        p.W = small ? 4 : 8;
        p.r2 = false;
        return p;
    }
    if (allow_split && force_s > 1 && nst / force_s >= 2 && K >= 1536 && tiles128 <= 256) p.splitk = force_s;
    if (p.splitk > 1 && !p.ws) p.BM = 128;
    if (p.BM == 128 && w8 && !p.ws) { p.W = 8; p.r2 = true; }
    return p;
}

int gn_run(const void* a, const void* b, const float* bias, const void* aux, void* c, void* c2, int64_t M, int32_t N,
           int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t epilogue, void* work, int64_t work_bytes, void* stream) {
    if (!a || !b || !c || M <= 0 || N <= 0 || K <= 0) return CLV_ERR_ARG;
    if (!clv_gemm_nt_supported(M, N, K)) return CLV_ERR_UNSUPPORTED;
    if ((lda & 7) || (ldb & 7) || (ldc & 7) || lda < K || ldb < K || ldc < N) return CLV_ERR_ARG;
    if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)c2) | ((uintptr_t)aux) | ((uintptr_t)bias) |
         ((uintptr_t)work)) & 15)
        return CLV_ERR_ARG;
    if ((epilogue == GN_EPI_BIAS || epilogue == GN_EPI_GELU || epilogue == GN_EPI_GELUD) && !bias) return CLV_ERR_ARG;
    if ((epilogue == GN_EPI_GELU || epilogue == GN_EPI_GELUD) && !c2) return CLV_ERR_ARG;
    if ((epilogue == GN_EPI_DGELU || epilogue == GN_EPI_MUL) && !aux) return CLV_ERR_ARG;
    if (epilogue < GN_EPI_NONE || epilogue > GN_EPI_MUL) return CLV_ERR_UNSUPPORTED;
    const char* force = getenv("CLV_GEMM_TILE");             // probe override: "128x128", "256x128", "128x128w8"
    if (force && (!strcmp(force, "lean") || !strcmp(force, "lean64"))) {     // one tile per workgroup (measured 5-15 % behind)
        // 4 GiB of addressable operand per tile row block (32-bit lane offsets)
        if ((int64_t)128 * lda * 2 >= (1ll << 31) || (int64_t)128 * ldb * 2 >= (1ll << 31)) return CLV_ERR_UNSUPPORTED;
        const int BNl = ((force && !strcmp(force, "lean64")) || (N % 128 != 0 && N <= 640)) ? 64 : 128;
        const int tilesN = (N + BNl - 1) / BNl;
        const int nmblk = (int)((M + 127) / 128);
        const unsigned grid = (unsigned)(8 * tilesN * ((nmblk + 7) / 8));
        hipStream_t st = (hipStream_t)stream;
        if (BNl == 128)
            return gn_launch_lean<128>(epilogue, st, grid, (const bf16_t*)a, (const bf16_t*)b, bias, (const bf16_t*)aux,
                                       (bf16_t*)c, (bf16_t*)c2, M, N, K, lda, ldb, ldc, tilesN, nmblk);
        return gn_launch_lean<64>(epilogue, st, grid, (const bf16_t*)a, (const bf16_t*)b, bias, (const bf16_t*)aux,
                                  (bf16_t*)c, (bf16_t*)c2, M, N, K, lda, ldb, ldc, tilesN, nmblk);
    }
    // few tiles and a long contraction (Swin stage 3 proj / fc2 / merge, fusion and text encoders: <= 384 tiles of
    // 128 x 128 for 512 workgroup slots): 64 x 128 tiles double the workgroups that share the DMA latency
    // (tools/probes/gemm_tiles.py; with the XCD column split the 450-tile qkv of stage 3 is faster on 128 x 128).
    // 128 x 128 tiles run on EIGHT waves (64 x 32 each) with a ring of 2: two workgroups = 16 waves per CU instead of 8 —
    // 0-5 % on every shape of the class (gemm_tiles.py: fc1 s3 24.8 -> 23.6, fc1 s1 29.5 -> 27.9, dqkv s2 20.2 -> 19.7 us)
    GnPlan pl = gn_plan(M, N, K, work != nullptr);
    if (pl.splitk > 1 && work_bytes < (int64_t)pl.splitk * M * N * 4) pl = gn_plan(M, N, K, false);
    int BM = pl.BM, BN = pl.BN, W = pl.W;
    bool r2 = pl.r2;
    if (force && !strncmp(force, "ws", 2)) {                  // wave-specialised classes (probe names: ws128, ws128c8, ws64, ws64r3)
        const int BMw = strstr(force, "64") ? 64 : 128;
        const int tilesN = (N + 127) / 128;
        const int nmblk = (int)((M + BMw - 1) / BMw);
        const bool two = !strcmp(force, "ws64r3");            // 72 KiB ring: two workgroups per CU
        const int pc = gn_pick_pc((int64_t)M * K * 2, (int64_t)N * K * 2, tilesN);
        const int max_units_xcd = ((nmblk + 8 / pc - 1) / (8 / pc)) * ((tilesN + pc - 1) / pc) * pl.splitk;
        const int cap = 32 * (two ? 2 : 1);
        const unsigned grid = (unsigned)(8 * (max_units_xcd < cap ? max_units_xcd : cap));
        hipStream_t st = (hipStream_t)stream;
#define GN_WARGS epilogue, st, grid, (const bf16_t*)a, (const bf16_t*)b, bias, (const bf16_t*)aux, (bf16_t*)c, (bf16_t*)c2, M, N, K, \
                 lda, ldb, ldc, tilesN, nmblk, pc, pl.splitk, (float*)work
        if (!strcmp(force, "ws128")) return gn_launch_ws<128, 128, 2, 2, 2, 4>(GN_WARGS);     // 4 consumers (64 x 64) + 2 producers
        if (!strcmp(force, "ws128c8")) return gn_launch_ws<128, 128, 2, 4, 2, 4>(GN_WARGS);   // 8 consumers (64 x 32) + 2 producers
        if (!strcmp(force, "ws64")) return gn_launch_ws<64, 128, 2, 2, 1, 4>(GN_WARGS);       // 4 consumers (32 x 64) + 1 producer, ring of 4
        if (!strcmp(force, "ws64p2")) return gn_launch_ws<64, 128, 2, 2, 2, 4>(GN_WARGS);
        if (!strcmp(force, "ws64r3")) return gn_launch_ws<64, 128, 2, 2, 1, 3>(GN_WARGS);     // ring of 3: two workgroups per CU
#undef GN_WARGS
        return CLV_ERR_UNSUPPORTED;
    }
    if (force) {
        BM = atoi(force);
        const char* x = strchr(force, 'x');
        if (x) BN = atoi(x + 1);
        const char* w = strchr(force, 'w');
        W = w ? atoi(w + 1) : (BM == 256 ? 8 : 4);
        r2 = strstr(force, "r2") != nullptr;
    }
    const int tilesN = (N + BN - 1) / BN;
    const int nmblk = (int)((M + BM - 1) / BM);
    if (!force && pl.ws) {                                    // wave-specialised classes: one workgroup per CU
        const int pcw = gn_pick_pc((int64_t)M * K * 2, (int64_t)N * K * 2, tilesN);
        const int max_units = ((nmblk + 8 / pcw - 1) / (8 / pcw)) * ((tilesN + pcw - 1) / pcw) * pl.splitk;
        const unsigned gridw = (unsigned)(8 * (max_units < 32 ? max_units : 32));
        hipStream_t stw = (hipStream_t)stream;
#define GN_WARGS epilogue, stw, gridw, (const bf16_t*)a, (const bf16_t*)b, bias, (const bf16_t*)aux, (bf16_t*)c, (bf16_t*)c2, M, N, \
                 K, lda, ldb, ldc, tilesN, nmblk, pcw, pl.splitk, (float*)work
        if (pl.ws == 1) return gn_launch_ws<128, 128, 2, 4, 2, 4>(GN_WARGS);
        return gn_launch_ws<64, 128, 2, 2, 2, 4>(GN_WARGS);
#undef GN_WARGS
    }
    // persistent workgroups: per_cu per CU, 32 * per_cu slots per XCD, never more than the fullest XCD's units
    const int per_cu = (BM == 128 && W == 4) ? 2 : (BM == 64 && (BN == 128 || BN == 192)) ? 2 : (BM == 64 && BN == 64) ? 3 : (BM == 128 && W == 8 && r2) ? 2 : 1;
    const int pc = gn_pick_pc((int64_t)M * K * 2, (int64_t)N * K * 2, tilesN);
    const int max_units_xcd = ((nmblk + 8 / pc - 1) / (8 / pc)) * ((tilesN + pc - 1) / pc) * pl.splitk;
    const unsigned grid = (unsigned)(8 * (max_units_xcd < 32 * per_cu ? max_units_xcd : 32 * per_cu));
    hipStream_t st = (hipStream_t)stream;
#define GN_ARGS epilogue, st, grid, (const bf16_t*)a, (const bf16_t*)b, bias, (const bf16_t*)aux, (bf16_t*)c, (bf16_t*)c2, M, N, K, \
                lda, ldb, ldc, tilesN, nmblk, pc, pl.splitk, pl.rot, (float*)work
    if (BM == 128 && BN == 128 && W == 4) return gn_launch<128, 128, 2, 2, 2>(GN_ARGS);   // 32 KiB stages x 2, two WGs per CU
    if (BM == 256 && BN == 128 && W == 8) return gn_launch<256, 128, 4, 2, 3>(GN_ARGS);   // 48 KiB stages x 3
    if (BM == 128 && BN == 128 && W == 8 && r2) return gn_launch<128, 128, 2, 4, 2>(GN_ARGS);   // 32 KiB stages x 2, two WGs per CU
    if (BM == 128 && BN == 128 && W == 8) return gn_launch<128, 128, 2, 4, 4>(GN_ARGS);   // 32 KiB stages x 4
    if (BM == 64 && BN == 128 && W == 4) return gn_launch<64, 128, 2, 2, 3>(GN_ARGS);     // 24 KiB stages x 3, two WGs per CU
    if (BM == 64 && BN == 128 && W == 8) return gn_launch<64, 128, 2, 4, 3>(GN_ARGS);     // the same ring on eight waves of 32 x 32
    if (BM == 64 && BN == 64 && W == 2) return gn_launch<64, 64, 1, 2, 3>(GN_ARGS);       // 16 KiB stages x 3, three WGs per CU
    if (BM == 128 && BN == 192 && W == 8) return gn_launch<128, 192, 4, 2, 3>(GN_ARGS);   // 40 KiB stages x 3, eight waves of 32 x 96
    if (BM == 64 && BN == 192 && W == 4) return gn_launch<64, 192, 2, 2, 2>(GN_ARGS);     // 32 KiB stages x 2, four waves of 32 x 96, two WGs per CU
#undef GN_ARGS
    return CLV_ERR_UNSUPPORTED;
}

}  // namespace

extern "C" int clv_gemm_nt(const void* a, const void* b, const float* bias, const void* aux, void* c, void* c2, int64_t M,
                           int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t epilogue, void* stream) {
    return gn_run(a, b, bias, aux, c, c2, M, N, K, lda, ldb, ldc, epilogue, nullptr, 0, stream);
}

extern "C" int64_t clv_gemm_nt_work_bytes(int64_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0 || !clv_gemm_nt_supported(M, N, K)) return 0;
    const GnPlan pl = gn_plan(M, N, K, true);
    return pl.splitk > 1 ? (int64_t)pl.splitk * M * N * 4 : 0;
}

extern "C" int clv_gemm_nt_ex(const void* a, const void* b, const float* bias, const void* aux, void* c, void* c2, int64_t M,
                              int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t epilogue, void* work,
                              int64_t work_bytes, void* stream) {
    return gn_run(a, b, bias, aux, c, c2, M, N, K, lda, ldb, ldc, epilogue, work, work_bytes, stream);
}

extern "C" int clv_transpose_batch(const void* src_base, void* dst_base, const void* table, int32_t n_entries,
                                   int32_t total_tiles, void* stream) {
    if (!src_base || !dst_base || !table || n_entries <= 0 || total_tiles <= 0) return CLV_ERR_ARG;
    static_assert(sizeof(TrEntry) == CLV_TRANSPOSE_ENTRY_BYTES, "TrEntry layout is part of the ABI");
    hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src_base, (bf16_t*)dst_base, (const TrEntry*)table, (int)n_entries);
    return clv_check_launch();
}
