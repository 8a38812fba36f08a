// Fused multi-head attention for the Clover path (gfx950, bf16 MFMA 16x16x32).
//
//   mode 1: WindowAttention3D (reference swin_transformer_3d.py:375-400) with the
//           cyclic shift + window partition/reverse (:459-476) folded into the
//           load/store index math — q/k/v/o live in the natural [B,D,H,W,*] layout.
//   mode 0: BERT self-attention over a [B,S,*] sequence with an additive key mask.
//
// One workgroup (4 waves) = one (group, head).  K (row-major) and V (transposed)
// are staged in LDS; each wave owns 16-query tiles.  QKᵀ is computed SWAPPED
// (Sᵀ = K·Qᵀ) so that a lane holds, for ONE query (lane&15), the scores of keys
// t*16 + (lane>>4)*4 + r — a row softmax is then lane-local plus two xor-shuffles,
// and the bf16 P fragment feeds the PV MFMA (Oᵀ = Vᵀ·Pᵀ) without any transpose:
// the MFMA k-index κ = (lane>>4)*8 + j simply enumerates the keys
// {2s*16 + (lane>>4)*4 + (j&3), j<4} ∪ {(2s+1)*16 + ...} and Vᵀ is read with the
// same enumeration.
#include "common.hpp"
#include <type_traits>
#include "../../include/clover_hip.h"

namespace {

constexpr int WAVES = 4;
constexpr int THREADS = WAVES * 64;
// The backward kernels run 8 waves per workgroup from 14 key tiles up: twice as many waves share the staged K / V
// (Q / dO), which is what limits the workgroups per CU (392-token windows: 72 KB -> two workgroups, -40 % per
// kernel; 196-token windows: 50 KB -> three, -7 % / -13 %)
constexpr int DKV_THREADS(int nkt) { return nkt >= 13 ? 512 : THREADS; }

struct Geom {
    ClvAttnGeom g;
    int nWh, nWw, nW;  // windows per axis / per clip (mode 1)
    unsigned drop_thresh;   // attention-probability dropout: P(drop) = drop_thresh / 2^32 (0 = off)
    float inv_keep;
    // relative-position bias table (mode 1): bias[i][j] = table[h][lin(i) - lin(j) + tcst] with
    // lin(n) = (n / (bwh*bww)) * ts_d + ((n / bww) % bwh) * ts_h + n % bww  — the reference's
    // relative_position_index[:N,:N] of the window the table was built for (swin_transformer_3d.py:343-360,386)
    int ts_d, ts_h, tcst, tlen, tls;     // tls: LDS floats of the table band, rounded up to 4
    int tb0, tbn;                        // the band [tb0, tb0 + tbn) of table rows a window of depth wd <= bwd can reach;
                                         // tcst is relative to tb0 (LDS addressing), tlen the full row count
    // few (group, head) pairs (the fusion encoder: 16 x 12 = 192 on 256 CUs): tsplit workgroups per pair, each staging
    // the pair's K / V (or Q / dO) and taking every tsplit-th set of 4 query (key) tiles
    int tsplit;
    // long sequences (mode 0, 448 < N <= 896 keys): the tokens STAGED in LDS — keys in the forward / dQ kernels, queries in
    // the dK / dV kernel — are split into nparts contiguous parts of pt16 tokens, one workgroup set per part; each set
    // writes its partial result (o + lse per key part, dq per key part, dk / dv per query part) to the caller's scratch
    // and a combine kernel merges them (seq_combine_*).  nparts = 1: everything below degenerates to the plain kernels.
    int grp0;                            // first group of this launch (the table-gradient path runs the dQ kernel in chunks)
    int nparts, pt16;
    int lddq, lddk, lddv;                // row strides of the dq / dk / dv OUTPUTS (= ldq / ldk / ldv unless partial)
    int64_t o_ps, lse_ps, dq_ps, dk_ps, dv_ps;   // element strides between the parts' partial outputs (0: one part)
};

__device__ __forceinline__ int win_lin(const Geom& G, int n) {
    const int hw = G.g.bwh * G.g.bww;
    const int tz = n / hw, tr = n - tz * hw;
    const int ty = tr / G.g.bww, tx = tr - ty * G.g.bww;
    return tz * G.ts_d + ty * G.ts_h + tx;
}

// Per-workgroup bias state in LDS (mode 1): the head's table and 4 * lin(n) per window token (byte offsets).
template <int NK>
__device__ __forceinline__ void load_bias_table(const Geom& G, const float* table, int h, float* tab_s, int* linb_s,
                                                int tid, int nthreads, float mul = 1.f) {
    for (int i = tid; i < G.tbn; i += nthreads) tab_s[i] = table[(int64_t)(G.tb0 + i) * G.g.nH + h] * mul;   // [tlen][nH] parameter layout
    for (int n = tid; n < NK; n += nthreads) linb_s[n] = (n < G.g.N) ? 4 * win_lin(G, n) : 0;
}

__device__ __forceinline__ int64_t tok_row(const Geom& G, int grp, int n) {
    if (G.g.mode == 0) return (int64_t)grp * G.g.N + n;
    const int b = grp / G.nW, wi = grp - b * G.nW;
    const int wz = wi / (G.nWh * G.nWw), wr = wi - wz * (G.nWh * G.nWw);
    const int wy = wr / G.nWw, wx = wr - wy * G.nWw;
    const int tz = n / (G.g.wh * G.g.ww), tr = n - tz * (G.g.wh * G.g.ww);
    const int ty = tr / G.g.ww, tx = tr - ty * G.g.ww;
    int d = wz * G.g.wd + tz + G.g.sd; if (d >= G.g.D) d -= G.g.D;   // roll(-shift): shifted[d'] = x[(d'+s) mod D]
    int h = wy * G.g.wh + ty + G.g.sh; if (h >= G.g.H) h -= G.g.H;
    int w = wx * G.g.ww + tx + G.g.sw; if (w >= G.g.W) w -= G.g.W;
    return (((int64_t)b * G.g.D + d) * G.g.H + h) * G.g.W + w;
}

// bias of 4 consecutive keys for one query: tp = table + 4 * (lin(q) + tcst) bytes, kb = 4 * lin(key..key+3)
__device__ __forceinline__ float4 table_bias4(const char* tp, const int* kb_ptr) {
    const int4 kb = *reinterpret_cast<const int4*>(kb_ptr);
    float4 b;
    b.x = *reinterpret_cast<const float*>(tp - kb.x);
    b.y = *reinterpret_cast<const float*>(tp - kb.y);
    b.z = *reinterpret_cast<const float*>(tp - kb.z);
    b.w = *reinterpret_cast<const float*>(tp - kb.w);
    return b;
}

// XCD-aware remap: consecutive logical ids (the heads of one window) share an XCD/L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int HD>
__device__ __forceinline__ void load_frags(Frag8 (&f)[(HD + 31) / 32], const bf16_t* rowp, bool valid, int lane) {
    constexpr int KS = (HD + 31) / 32;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int hd0 = s * 32 + (lane >> 4) * 8;
        if (valid && hd0 < HD) f[s].u4 = *reinterpret_cast<const uint4*>(rowp + hd0);
        else f[s].u4 = make_uint4(0, 0, 0, 0);
    }
}

template <int HD>
__device__ __forceinline__ void lds_frags(Frag8 (&f)[(HD + 31) / 32], const bf16_t* rowp, int lane) {
    constexpr int KS = (HD + 31) / 32;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int hd0 = s * 32 + (lane >> 4) * 8;
        if (hd0 < HD) f[s].u4 = *reinterpret_cast<const uint4*>(rowp + hd0);
        else f[s].u4 = make_uint4(0, 0, 0, 0);
    }
}

// Stage `N` rows (tensor rows row_s[n]) of a [tokens][ld] bf16 matrix (cols h*HD .. +HD) row-major into
// LDS rm[NK][HD+8]; pad rows are zeroed.  Transposed operands are NOT materialised: they are read
// with the gfx950 LDS transpose read (tr4 below).
template <int HD, int NK, bool SCALE = false>
__device__ __forceinline__ void stage(const int* row_s, const bf16_t* base, int ld, int h, int N, bf16_t* rm, int tid,
                                      int nthr = THREADS, float mul = 1.f) {
    constexpr int CH = HD / 8, LDR = HD + 8;
    for (int idx = tid; idx < NK * CH; idx += nthr) {
        const int n = idx / CH, c = idx - n * CH;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (n < N) val = *reinterpret_cast<const uint4*>(base + (int64_t)row_s[n] * ld + h * HD + c * 8);
        if (SCALE) {                 // fold softmax scale * log2(e) into the staged operand: the scores leave the MFMA ready for exp2
            uint32_t* w = reinterpret_cast<uint32_t*>(&val);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = pack2bf(half_lo(w[e]) * mul, half_hi(w[e]) * mul);
        }
        *reinterpret_cast<uint4*>(rm + n * LDR + c * 8) = val;
    }
}

// The token -> tensor-row map of this window (roll / partition folded in, ~60 integer VALU ops with five
// divisions) is evaluated ONCE per workgroup into LDS; staging and the per-tile operand loads index it.
template <int NK>
__device__ __forceinline__ void token_rows(const Geom& G, int grp, int* row_s, int tid, int nthr = THREADS, int s0 = 0) {
    for (int n = tid; n < NK; n += nthr) row_s[n] = (s0 + n < G.g.N) ? (int)tok_row(G, grp, s0 + n) : 0;
    __syncthreads();
}
// tensor row of a LOOPED token (query in fwd / dQ, key in dK / dV): the LDS map covers the staged part only once a
// sequence is split, and a sequence's map is one multiply-add anyway
__device__ __forceinline__ int64_t loop_row(const Geom& G, const int* row_s, int grp, int n) {
    return G.nparts > 1 ? (int64_t)grp * G.g.N + n : (int64_t)row_s[n];
}

typedef short v4s_t __attribute__((ext_vector_type(4)));
// ds_read_b64_tr_b16: from a row-major LDS matrix (leading dim LD elements) return, for this lane,
// M[row0 + 0..3][c0 + (lane&15)] — i.e. 4 consecutive ROWS of one column, the k-contiguous half of an
// MFMA operand whose k index runs along the rows.  Lane i of each 16-lane group addresses the i-th
// 8-byte piece of the 4x16 block (row i>>2, cols (i&3)*4..+3); the hardware transposes the block.
__device__ __forceinline__ uint2 tr4(const bf16_t* base, int LD, int row0, int c0, int lr) {
    const bf16_t* p = base + (row0 + (lr >> 2)) * LD + c0 + (lr & 3) * 4;
    const v4s_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)p);
    union { v4s_t v; uint2 u; } cv;
    cv.v = r;
    return cv.u;
}

// exp(x - L) as ONE v_exp_f32: exp2(x * log2e - L * log2e).  __expf costs a multiply plus two v_cndmask of
// denormal-range handling per element, which is a third of the softmax VALU work here; a result below 2^-126 is
// flushed to zero, which is what a probability that small contributes anyway.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float exp_sub(float x, float negL2) { return __builtin_amdgcn_exp2f(fmaf(x, LOG2E, negL2)); }

// ------------------------------------------------------------------------- forward
template <int HD, int NKT, bool DROP, int MODE>
__global__ void __launch_bounds__(THREADS, (NKT <= 16 ? 3 : 1)) attn_fwd_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    bf16_t* __restrict__ o, float* __restrict__ lse, const float* __restrict__ bias,
    const int* __restrict__ rid, const float* __restrict__ kmask, const unsigned long long* __restrict__ seedp,
    Geom G) {
    constexpr int NK = NKT * 16, KS = (HD + 31) / 32, LDR = HD + 8, NC = HD / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NK * LDR;
    float* aux = reinterpret_cast<float*>(Vs + NK * LDR);   // NK floats: kmask (mode 0) / rid as int (mode 1)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int part = bid % G.tsplit;                          // tsplit workgroups share one (group, head): tile subsets
    const int sp = (bid / G.tsplit) % G.nparts, gh = bid / (G.tsplit * G.nparts);     // sp: which part of the keys is staged
    const int grp = gh / G.g.nH, h = gh - grp * G.g.nH;
    const int N = G.g.N;
    const int s0 = sp * G.pt16, Ns = min(N - s0, G.pt16);     // staged keys [s0, s0 + Ns); one part: all N

    // window mode keeps no additive key array (its only term is -inf on the pad keys of the last tile: computed): with
    // 13 key tiles the workgroup then needs < 40 KB of LDS and FOUR share a CU instead of three
    int* row_s = reinterpret_cast<int*>(aux + (MODE == 1 ? 1 : 2) * NK);
    int* linb_s = row_s + NK;                                // mode 1 + bias: 4 * lin(n)
    float* tab_s = reinterpret_cast<float*>(linb_s + NK);    //                the head's bias table
    int* flag_s = reinterpret_cast<int*>(tab_s + G.tls);     // "window straddles shift regions"
    if (tid == 0) *flag_s = 0;
    token_rows<NK>(G, grp, row_s, tid, THREADS, s0);
    // Q fragments of ALL this wave's query tiles go out before the K / V staging loads: their round trip (~2 us under
    // load) then overlaps the staging instead of stalling every tile of the loop below.  (Long windows keep the
    // per-tile load: their score registers leave no room.)
    constexpr int PRE = NKT <= 16 ? (NKT + WAVES - 1) / WAVES : 1;
    const int nqt = (N + 15) >> 4;
    Frag8 qpre[PRE][KS];
    if (PRE > 1) {
#pragma unroll
        for (int ti = 0; ti < PRE; ++ti) {
            const int qt = wave + WAVES * part + ti * WAVES * G.tsplit;
            const int nq = qt * 16 + (lane & 15);
            const bool qv = qt < nqt && nq < N;
            load_frags<HD>(qpre[ti], q + loop_row(G, row_s, grp, qv ? nq : 0) * G.g.ldq + h * HD, qv, lane);
        }
    }
    // K is staged as K * scale * log2(e) and every additive term (bias table, key mask, region mask) is kept in log2
    // units: the MFMA, started from the additive terms as its accumulator, delivers the exp2-ready score — one VALU
    // operation per score less than scale-and-add after the MFMA.
    stage<HD, NK, true>(row_s, k, G.g.ldk, h, Ns, Ks, tid, THREADS, G.g.scale * LOG2E);
    stage<HD, NK>(row_s, v, G.g.ldv, h, Ns, Vs, tid);
    if (MODE == 1 && bias) load_bias_table<NK>(G, bias, h, tab_s, linb_s, tid, THREADS, LOG2E);
    int* rid_s = reinterpret_cast<int*>(aux);
    // kadd[n]: additive key term — the key mask (mode 0) for real keys, -inf for the pad keys of the last tile.
    float* kadd = aux + NK;
    const int wloc = (G.g.mode == 1) ? grp % G.nW : 0;
    int differs = 0;
    for (int n = tid; n < NK; n += THREADS) {
        if (MODE == 1 && rid) {
            const int rv = (n < N) ? rid[wloc * N + n] : 0;
            rid_s[n] = rv;
            if (n < N) differs |= rv != rid[wloc * N];
        }
        if (MODE == 0) kadd[n] = (n < Ns) ? (kmask ? kmask[(int64_t)grp * N + s0 + n] * LOG2E : 0.f) : -INFINITY;
    }
    // a shifted block's windows that lie inside ONE region (all but the last row / column of windows: 49 of 64 at
    // 56 x 56) need no mask: 12 VALU operations per key tile and query less
    if (differs) *flag_s = 1;
    __syncthreads();
    const bool masked = *flag_s != 0;

    const int lg = lane >> 4, lr = lane & 15;
    Frag8 onesf;
    onesf.u[0] = onesf.u[1] = onesf.u[2] = onesf.u[3] = CLV_ONE_PAIR;      // 1.0 pairs
    constexpr int UNR = PRE > 1 ? PRE : 1;
#pragma unroll UNR
    for (int ti = 0; ti < (NKT + WAVES - 1) / WAVES; ++ti) {
        const int qt = wave + WAVES * part + ti * WAVES * G.tsplit;
        if (qt >= nqt) break;
        const int nq = qt * 16 + lr;
        const bool qv = nq < N;
        const int64_t qrow = loop_row(G, row_s, grp, qv ? nq : 0);
        Frag8 qf[KS];
        if (PRE > 1) {
#pragma unroll
            for (int s = 0; s < KS; ++s) qf[s] = qpre[PRE > 1 ? ti : 0][s];
        } else {
            load_frags<HD>(qf, q + qrow * G.g.ldq + h * HD, qv, lane);
        }

        constexpr int CH = NKT > 16 ? 14 : NKT;             // key tiles per softmax chunk
        float p[CH][4];
        // relative-position bias from the LDS table: byte offset = 4 * (lin(q) + tcst) - 4 * lin(key)
        const bool tb = MODE == 1 && bias != nullptr;
        const char* tp = reinterpret_cast<const char*>(tab_s) + (tb ? linb_s[qv ? nq : 0] + 4 * G.tcst : 0);
        const int rq = (MODE == 1 && rid && qv) ? rid_s[nq] : 0;
        float4 bnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tb) bnext = table_bias4(tp, linb_s + lg * 4);
        // Long windows / sequences (25 / 28 key tiles) walk their keys in TWO chunks of <= 14 tiles with a running row
        // maximum (the accumulators and the denominator are rescaled by exp2(m_old - m_new) between them): 56 score
        // registers instead of 100-112, so that the kernel fits 168 VGPRs = 3 waves per SIMD and two workgroups share a CU
        // (one workgroup of one wave per SIMD before: 260 us per stage-0 block of Swin-B at 16 frames).
        f32x4_t oacc[NC];
        f32x4_t sacc = {0.f, 0.f, 0.f, 0.f};                 // row sums of P from the matrix pipe (all-ones "V" block)
#pragma unroll
        for (int c = 0; c < NC; ++c) oacc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        float m = -INFINITY, sum = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < NKT; c0 += CH) {
            constexpr int dummy = 0; (void)dummy;
            const int ct = (NKT - c0) < CH ? (NKT - c0) : CH;  // tiles of this chunk (compile-time after unrolling)
            float mc = -INFINITY;
#pragma unroll
            for (int tt = 0; tt < CH; ++tt) {
                if (tt >= ct) break;
                const int t = c0 + tt;
                Frag8 kf[KS];
                lds_frags<HD>(kf, Ks + (t * 16 + lr) * LDR, lane);
                const int key0 = t * 16 + lg * 4;
                asm volatile("" ::: "memory");               // keep the gathers one tile ahead, not all up front
                const float4 bv = bnext;
                if (tb && t + 1 < NKT) bnext = table_bias4(tp, linb_s + (t + 1) * 16 + lg * 4);
                f32x4_t acc = {bv.x, bv.y, bv.z, bv.w};               // additive terms first, the MFMA adds q.k on top
                if (MODE == 0) {                                      // key mask, -inf on the pad keys of the last tile
                    const float4 ka = *reinterpret_cast<const float4*>(kadd + key0);
                    acc[0] += ka.x; acc[1] += ka.y; acc[2] += ka.z; acc[3] += ka.w;
                } else if ((t + 1) * 16 > Ns) {                       // window mode: only the pad keys of the last tile(s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = (key0 + r < Ns) ? acc[r] : -INFINITY;
                }
                if (masked) {                                         // workgroup-uniform: window straddles shift regions
                    const int4 rk = *reinterpret_cast<const int4*>(rid_s + key0);
                    acc[0] += (rk.x != rq) ? -100.0f * LOG2E : 0.0f;
                    acc[1] += (rk.y != rq) ? -100.0f * LOG2E : 0.0f;
                    acc[2] += (rk.z != rq) ? -100.0f * LOG2E : 0.0f;
                    acc[3] += (rk.w != rq) ? -100.0f * LOG2E : 0.0f;
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) acc = mfma16(kf[s], qf[s], acc);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[tt][r] = acc[r];
                    mc = fmaxf(mc, acc[r]);
                }
            }
            mc = grp4_max(mc);                               // log2 units
            if (CH < NKT) {                                  // several chunks: running maximum, rescale what is accumulated
                const float mn = fmaxf(m, mc);
                const float sc = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
                if (c0 > 0) {
#pragma unroll
                    for (int c = 0; c < NC; ++c)
#pragma unroll
                        for (int r = 0; r < 4; ++r) oacc[c][r] *= sc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) sacc[r] *= sc;
                    sum *= sc;
                }
                m = mn;
            } else {
                m = mc;
            }
            float csum = 0.f;
#pragma unroll
            for (int tt = 0; tt < CH; ++tt) {
                if (tt >= ct) break;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(p[tt][r] - m);
                    p[tt][r] = e;
                    if (DROP) csum += e;
                }
            }
            if (DROP) {                                    // dropout on the probabilities (after softmax)
                sum += grp4_sum(csum);
                const unsigned long long sd = *seedp;
                const unsigned rowid = (unsigned)((grp * G.g.nH + h) * N + nq);
#pragma unroll
                for (int tt = 0; tt < CH; ++tt) {
                    if (tt >= ct) break;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        p[tt][r] *= keep_scale(sd, rowid, (unsigned)(s0 + (c0 + tt) * 16 + lg * 4 + r), G.drop_thresh, G.inv_keep);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < (CH + 1) / 2; ++s2) {
                if (2 * s2 >= ct) break;
                const bool tail = 2 * s2 + 1 >= ct;          // odd tile count of the (last) chunk: half-empty k-step
                Frag8 pf;
                pf.u[0] = pack2bf(p[2 * s2][0], p[2 * s2][1]);
                pf.u[1] = pack2bf(p[2 * s2][2], p[2 * s2][3]);
                pf.u[2] = tail ? 0u : pack2bf(p[tail ? 0 : 2 * s2 + 1][0], p[tail ? 0 : 2 * s2 + 1][1]);
                pf.u[3] = tail ? 0u : pack2bf(p[tail ? 0 : 2 * s2 + 1][2], p[tail ? 0 : 2 * s2 + 1][3]);
                const int kt = c0 + 2 * s2;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    Frag8 vf;   // A[hd c*16+lr][kappa] = V[key(kappa)][hd]
                    vf.u2[0] = tr4(Vs, LDR, kt * 16 + lg * 4, c * 16, lr);
                    vf.u2[1] = tail ? make_uint2(0u, 0u) : tr4(Vs, LDR, (kt + 1) * 16 + lg * 4, c * 16, lr);
                    oacc[c] = mfma16(vf, pf, oacc[c]);
                }
                // without dropout the softmax denominator is the sum of the SAME bf16 probabilities the P.V product uses:
                // seven idle-pipe MFMAs instead of 56 adds and two shuffles per query tile
                if (!DROP) sacc = mfma16(onesf, pf, sacc);
            }
        }
        if (!DROP) sum = sacc[0];
        // one part: the final o / lse.  Several: this key part's normalised o and its lse into the caller's scratch (part
        // stride o_ps / lse_ps), merged by seq_combine_fwd_kernel
        if (qv && lg == 0) lse[sp * G.lse_ps + ((int64_t)grp * G.g.nH + h) * N + nq] = (m + __log2f(sum)) * (1.0f / LOG2E);
        const float inv = 1.0f / sum;
        if (qv) {
            bf16_t* orow = o + sp * G.o_ps + qrow * G.g.ldo + h * HD + lg * 4;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uint2 w;
                w.x = pack2bf(oacc[c][0] * inv, oacc[c][1] * inv);
                w.y = pack2bf(oacc[c][2] * inv, oacc[c][3] * inv);
                *reinterpret_cast<uint2*>(orow + c * 16) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------- backward A: dQ, dS scratch, D
// Window mode also writes dS (bf16) of every (window, head) to a scratch; dbias_table_kernel sums it over the
// windows and scatters the sums into the table gradient.
template <int HD, int NKT, bool DROP, int MODE>
// 13-tile windows (8 waves): 80 VGPRs = 6 waves per SIMD = THREE workgroups per CU (the kernel wanted 82: two)
__global__ void __launch_bounds__(DKV_THREADS(NKT), (NKT > 16 ? 2 : ((NKT == 13 || NKT == 14) && HD == 32 && MODE == 1 ? 6 : 1))) attn_bwd_dq_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ bias, const int* __restrict__ rid, const float* __restrict__ kmask,
    bf16_t* __restrict__ dq, bf16_t* __restrict__ ds_out, float* __restrict__ dsum,
    const unsigned long long* __restrict__ seedp, Geom G) {
    constexpr int NK = NKT * 16, KS = (HD + 31) / 32, LDR = HD + 8, NC = HD / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
    bf16_t* Vs = Ks + NK * LDR;
    float* aux = reinterpret_cast<float*>(Vs + NK * LDR);
    int* rid_s = reinterpret_cast<int*>(aux);
    float* kadd = aux + NK;                                  // see attn_fwd_kernel
    int* row_s = reinterpret_cast<int*>(aux + 2 * NK);
    int* linb_s = row_s + NK;
    float* tab_s = reinterpret_cast<float*>(linb_s + NK);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int part = bid % G.tsplit;                          // tsplit workgroups share one (group, head): tile subsets
    const int sp = (bid / G.tsplit) % G.nparts, gh = bid / (G.tsplit * G.nparts);     // see attn_fwd_kernel
    const int grp = gh / G.g.nH + G.grp0, h = gh % G.g.nH;    // grp0: this launch covers groups [grp0, grp0 + gridDim / ..)
    const int N = G.g.N;
    const int s0 = sp * G.pt16, Ns = min(N - s0, G.pt16);     // staged keys [s0, s0 + Ns)
    const bool tb = MODE == 1 && bias != nullptr;
    int* flag_s = reinterpret_cast<int*>(tab_s + G.tls);
    if (tid == 0) *flag_s = 0;

    token_rows<NK>(G, grp, row_s, tid, nthr, s0);
    // q / dO / o fragments of all this wave's query tiles go out before the K / V staging loads (see attn_fwd_kernel)
    constexpr int NWAVES = DKV_THREADS(NKT) / 64;
    constexpr int PRE = (NKT + NWAVES - 1) / NWAVES;
    // long windows (25 / 28 tiles, 4 query tiles per wave) load the fragments per tile instead: 36 VGPRs that decide
    // between one and two 8-wave workgroups per CU (128-VGPR step)
    constexpr bool PREF = NKT <= 16;
    const int nqt = (N + 15) >> 4;
    Frag8 qpre[PREF ? PRE : 1][KS], dopre[PREF ? PRE : 1][KS], opre[PREF ? PRE : 1][KS];
    if (PREF) {
#pragma unroll
        for (int ti = 0; ti < PRE; ++ti) {
            const int qt = wave + nwaves * part + ti * nwaves * G.tsplit;
            const int nq = qt * 16 + (lane & 15);
            const bool qv = qt < nqt && nq < N;
            const int64_t qrow = loop_row(G, row_s, grp, qv ? nq : 0);
            load_frags<HD>(qpre[ti], q + qrow * G.g.ldq + h * HD, qv, lane);
            load_frags<HD>(dopre[ti], dout + qrow * G.g.ldo + h * HD, qv, lane);
            load_frags<HD>(opre[ti], o + qrow * G.g.ldo + h * HD, qv, lane);
        }
    }
    // K staged as K * scale * log2(e), additive terms in log2 units, MFMA started from them (see attn_fwd_kernel);
    // dQ = dS . K * scale then is (dS . K') / log2(e)
    stage<HD, NK, true>(row_s, k, G.g.ldk, h, Ns, Ks, tid, nthr, G.g.scale * LOG2E);
    stage<HD, NK>(row_s, v, G.g.ldv, h, Ns, Vs, tid, nthr);
    if (tb) load_bias_table<NK>(G, bias, h, tab_s, linb_s, tid, nthr, LOG2E);
    const int wloc = (G.g.mode == 1) ? grp % G.nW : 0;
    int differs = 0;
    for (int n = tid; n < NK; n += nthr) {
        if (MODE == 1 && rid) {
            const int rv = (n < N) ? rid[wloc * N + n] : 0;
            rid_s[n] = rv;
            if (n < N) differs |= rv != rid[wloc * N];
        }
        kadd[n] = (n < Ns) ? ((MODE == 0 && kmask) ? kmask[(int64_t)grp * N + s0 + n] * LOG2E : 0.f) : -INFINITY;
    }
    if (differs) *flag_s = 1;
    __syncthreads();
    const bool masked = *flag_s != 0;                        // see attn_fwd_kernel

    const int lg = lane >> 4, lr = lane & 15;
    const unsigned long long sd = DROP ? *seedp : 0ull;
#pragma unroll
    for (int ti = 0; ti < PRE; ++ti) {
        const int qt = wave + nwaves * part + ti * nwaves * G.tsplit;
        if (qt >= nqt) break;
        const int nq = qt * 16 + lr;
        const bool qv = nq < N;
        const int64_t qrow = loop_row(G, row_s, grp, qv ? nq : 0);
        Frag8 qf[KS], dof[KS], of[KS];
        if (PREF) {
#pragma unroll
            for (int s = 0; s < KS; ++s) { qf[s] = qpre[PREF ? ti : 0][s]; dof[s] = dopre[PREF ? ti : 0][s]; of[s] = opre[PREF ? ti : 0][s]; }
        } else {
            load_frags<HD>(qf, q + qrow * G.g.ldq + h * HD, qv, lane);
            load_frags<HD>(dof, dout + qrow * G.g.ldo + h * HD, qv, lane);
            load_frags<HD>(of, o + qrow * G.g.ldo + h * HD, qv, lane);
        }
        float dsm = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) dsm += bf2f(dof[s].h[e]) * bf2f(of[s].h[e]);
        dsm = grp4_sum(dsm);
        const int64_t li = ((int64_t)grp * G.g.nH + h) * N + nq;
        const unsigned rowid = (unsigned)li;
        if (qv && lg == 0) dsum[li] = dsm;
        const float nL2 = qv ? -lse[li] * LOG2E : 0.f;
        const char* tp = reinterpret_cast<const char*>(tab_s) + (tb ? linb_s[qv ? nq : 0] + 4 * G.tcst : 0);
        const int rq = (MODE == 1 && rid && qv) ? rid_s[nq] : 0;
        // dS scratch in MFMA-fragment order [group][head][q tile][key tile][lane][4]: every store instruction of a
        // wave is one contiguous 512-B run (row-major rows would be 32-B pieces of 128-B lines)
        bf16_t* dsfrag = (tb && ds_out)
            ? ds_out + ((((int64_t)(grp - G.grp0) * G.g.nH + h) * nqt + qt) * NKT * 64 + lane) * 4 : nullptr;

        // long windows walk their key tiles in chunks of 14: the packed dS of a chunk feeds the dQ MFMAs right away, so
        // 7 instead of 13-14 fragment registers stay live (with the per-tile fragment loads: 165 -> <= 128 VGPRs)
        constexpr int CH = NKT > 16 ? 14 : NKT;
        Frag8 dsf[(CH + 1) / 2];
        f32x4_t qacc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) qacc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        float4 bnext = make_float4(0.f, 0.f, 0.f, 0.f);
        if (tb) bnext = table_bias4(tp, linb_s + lg * 4);
#pragma unroll
        for (int c0 = 0; c0 < NKT; c0 += CH) {
            const int ct = (NKT - c0) < CH ? (NKT - c0) : CH;
            if (ct & 1) dsf[ct / 2].u[2] = dsf[ct / 2].u[3] = 0u;   // the missing second half of an odd tile count
#pragma unroll
            for (int tt = 0; tt < CH; ++tt) {
                if (tt >= ct) break;
                const int t = c0 + tt;
                asm volatile("" ::: "memory");               // keep the gathers one tile ahead, not all up front
                const float4 bv = bnext;
                if (tb && t + 1 < NKT) bnext = table_bias4(tp, linb_s + (t + 1) * 16 + lg * 4);
                f32x4_t sacc = {bv.x, bv.y, bv.z, bv.w}, pacc = {0.f, 0.f, 0.f, 0.f};
                const int key0 = t * 16 + lg * 4;
                if (MODE == 0 || (t + 1) * 16 > Ns) {
                    const float4 ka = *reinterpret_cast<const float4*>(kadd + key0);
                    sacc[0] += ka.x; sacc[1] += ka.y; sacc[2] += ka.z; sacc[3] += ka.w;
                }
                if (masked) {
                    const int4 rk = *reinterpret_cast<const int4*>(rid_s + key0);
                    sacc[0] += (rk.x != rq) ? -100.0f * LOG2E : 0.0f;
                    sacc[1] += (rk.y != rq) ? -100.0f * LOG2E : 0.0f;
                    sacc[2] += (rk.z != rq) ? -100.0f * LOG2E : 0.0f;
                    sacc[3] += (rk.w != rq) ? -100.0f * LOG2E : 0.0f;
                }
                Frag8 kf[KS], vf[KS];
                lds_frags<HD>(kf, Ks + (t * 16 + lr) * LDR, lane);
                lds_frags<HD>(vf, Vs + (t * 16 + lr) * LDR, lane);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    sacc = mfma16(kf[s], qf[s], sacc);
                    pacc = mfma16(vf[s], dof[s], pacc);
                }
                float ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pr = __builtin_amdgcn_exp2f(sacc[r] + nL2);           // pad keys: exp2(-inf) = 0
                    float dp = pacc[r];
                    if (DROP) dp *= keep_scale(sd, rowid, (unsigned)(s0 + key0 + r), G.drop_thresh, G.inv_keep);
                    ds[r] = pr * (dp - dsm);
                }
                dsf[tt >> 1].u[(tt & 1) * 2 + 0] = pack2bf(ds[0], ds[1]);
                dsf[tt >> 1].u[(tt & 1) * 2 + 1] = pack2bf(ds[2], ds[3]);
                if (dsfrag)
                    *reinterpret_cast<uint2*>(dsfrag + t * 256) =
                        make_uint2(dsf[tt >> 1].u[(tt & 1) * 2 + 0], dsf[tt >> 1].u[(tt & 1) * 2 + 1]);
            }
#pragma unroll
            for (int s2 = 0; s2 < (CH + 1) / 2; ++s2) {
                if (2 * s2 >= ct) break;
                const bool tail = 2 * s2 + 1 >= ct;
                const int kt = c0 + 2 * s2;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    Frag8 kf;   // A[hd c*16+lr][kappa] = K[key(kappa)][hd]
                    kf.u2[0] = tr4(Ks, LDR, kt * 16 + lg * 4, c * 16, lr);
                    kf.u2[1] = tail ? make_uint2(0u, 0u) : tr4(Ks, LDR, (kt + 1) * 16 + lg * 4, c * 16, lr);
                    qacc[c] = mfma16(kf, dsf[s2], qacc[c]);
                }
            }
        }
        if (qv) {
            bf16_t* drow = dq + sp * G.dq_ps + qrow * G.lddq + h * HD + lg * 4;    // several key parts: partial dq each
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uint2 w;
                w.x = pack2bf(qacc[c][0] * (1.0f / LOG2E), qacc[c][1] * (1.0f / LOG2E));
                w.y = pack2bf(qacc[c][2] * (1.0f / LOG2E), qacc[c][3] * (1.0f / LOG2E));
                *reinterpret_cast<uint2*>(drow + c * 16) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------- backward B: dK, dV
template <int HD, int NKT, bool DROP, int MODE>
__global__ void __launch_bounds__(DKV_THREADS(NKT)) attn_bwd_dkv_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ dout, const float* __restrict__ lse, const float* __restrict__ dsum,
    const float* __restrict__ bias, const int* __restrict__ rid, const float* __restrict__ kmask,
    bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, const unsigned long long* __restrict__ seedp, Geom G) {
    constexpr int NK = NKT * 16, KS = (HD + 31) / 32, LDR = HD + 8, NC = HD / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);
    bf16_t* dOs = Qs + NK * LDR;
    float* L_s = reinterpret_cast<float*>(dOs + NK * LDR);
    float* D_s = L_s + NK;
    float* aux = D_s + NK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int part = bid % G.tsplit;                          // tsplit workgroups share one (group, head): tile subsets
    const int sp = (bid / G.tsplit) % G.nparts, gh = bid / (G.tsplit * G.nparts);     // sp: which part of the QUERIES is staged
    const int grp = gh / G.g.nH, h = gh - grp * G.g.nH;
    const int N = G.g.N;
    const int s0 = sp * G.pt16, Ns = min(N - s0, G.pt16);     // staged queries [s0, s0 + Ns)

    int* row_s = reinterpret_cast<int*>(aux + NK);
    int* linb_s = row_s + NK;
    float* tab_s = reinterpret_cast<float*>(linb_s + NK);
    const bool tb = MODE == 1 && bias != nullptr;
    int* flag_s = reinterpret_cast<int*>(tab_s + G.tls);
    if (tid == 0) *flag_s = 0;
    token_rows<NK>(G, grp, row_s, tid, nthr, s0);
    // k / v fragments of all this wave's key tiles go out before the Q / dO staging loads (see attn_fwd_kernel)
    constexpr int NWAVES = DKV_THREADS(NKT) / 64;
    constexpr int PRE = (NKT + NWAVES - 1) / NWAVES;
    const int nkt = (N + 15) >> 4;
    Frag8 kpre[PRE][KS], vpre[PRE][KS];
#pragma unroll
    for (int ti = 0; ti < PRE; ++ti) {
        const int kt = wave + nwaves * part + ti * nwaves * G.tsplit;
        const int nk = kt * 16 + (lane & 15);
        const bool kv = kt < nkt && nk < N;
        const int64_t krow = loop_row(G, row_s, grp, kv ? nk : 0);
        load_frags<HD>(kpre[ti], k + krow * G.g.ldk + h * HD, kv, lane);
        load_frags<HD>(vpre[ti], v + krow * G.g.ldv + h * HD, kv, lane);
    }
    // Q staged as Q * scale * log2(e), additive terms in log2 units, MFMA started from them (see attn_fwd_kernel);
    // dK = dS^T . Q * scale then is (dS^T . Q') / log2(e)
    stage<HD, NK, true>(row_s, q, G.g.ldq, h, Ns, Qs, tid, nthr, G.g.scale * LOG2E);
    stage<HD, NK>(row_s, dout, G.g.ldo, h, Ns, dOs, tid, nthr);
    if (tb) load_bias_table<NK>(G, bias, h, tab_s, linb_s, tid, nthr, LOG2E);
    int* rid_s = reinterpret_cast<int*>(aux);
    const int wloc = (G.g.mode == 1) ? grp % G.nW : 0;
    int differs = 0;
    for (int n = tid; n < NK; n += nthr) {
        const int64_t li = ((int64_t)grp * G.g.nH + h) * N + s0 + n;
        L_s[n] = (n < Ns) ? -lse[li] * LOG2E : -INFINITY;   // -L * log2e (exp_sub's form); pad queries: P = exp2(-inf) = 0
        D_s[n] = (n < Ns) ? dsum[li] : 0.f;
        if (MODE == 1 && rid) {
            const int rv = (n < N) ? rid[wloc * N + n] : 0;
            rid_s[n] = rv;
            if (n < N) differs |= rv != rid[wloc * N];
        }
    }
    if (differs) *flag_s = 1;
    __syncthreads();
    const bool masked = *flag_s != 0;                        // see attn_fwd_kernel

    const int lg = lane >> 4, lr = lane & 15;
    const unsigned long long sd = DROP ? *seedp : 0ull;
#pragma unroll
    for (int ti = 0; ti < PRE; ++ti) {
        const int kt = wave + nwaves * part + ti * nwaves * G.tsplit;
        if (kt >= nkt) break;
        const int nk = kt * 16 + lr;
        const bool kv = nk < N;
        const int64_t krow = loop_row(G, row_s, grp, kv ? nk : 0);
        Frag8 kf[KS], vf[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { kf[s] = kpre[ti][s]; vf[s] = vpre[ti][s]; }
        const int rk = (MODE == 1 && rid && kv) ? rid_s[nk] : 0;
        const float kmv = (MODE == 0 && kmask && kv) ? kmask[(int64_t)grp * N + nk] * LOG2E : 0.f;   // this lane's key
        const int ko = tb ? linb_s[kv ? nk : 0] - 4 * G.tcst : 0;     // slot(q, key) = lin(q) - (lin(key) - tcst)

        f32x4_t dvacc[NC], dkacc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            dvacc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            dkacc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int qp = 0; qp < (NKT + 1) / 2; ++qp) {
            Frag8 pf, dsf;
            pf.u[2] = pf.u[3] = dsf.u[2] = dsf.u[3] = 0u;   // odd tile count: the last pair has no second query tile
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = qp * 2 + half;
                if ((NKT & 1) && qt >= NKT) break;
                float pv[4], dsv[4];
                const int qn0 = qt * 16 + lg * 4;
                float4 bv = make_float4(kmv, kmv, kmv, kmv);
                if (tb) {
                    const int4 qb = *reinterpret_cast<const int4*>(linb_s + qn0);
                    const char* tp = reinterpret_cast<const char*>(tab_s) - ko;
                    bv.x = *reinterpret_cast<const float*>(tp + qb.x);
                    bv.y = *reinterpret_cast<const float*>(tp + qb.y);
                    bv.z = *reinterpret_cast<const float*>(tp + qb.z);
                    bv.w = *reinterpret_cast<const float*>(tp + qb.w);
                }
                f32x4_t sacc = {bv.x, bv.y, bv.z, bv.w}, pacc = {0.f, 0.f, 0.f, 0.f};
                if (masked) {
                    const int4 rq4 = *reinterpret_cast<const int4*>(rid_s + qn0);
                    sacc[0] += (rq4.x != rk) ? -100.0f * LOG2E : 0.0f;
                    sacc[1] += (rq4.y != rk) ? -100.0f * LOG2E : 0.0f;
                    sacc[2] += (rq4.z != rk) ? -100.0f * LOG2E : 0.0f;
                    sacc[3] += (rq4.w != rk) ? -100.0f * LOG2E : 0.0f;
                }
                Frag8 qf[KS], dof[KS];
                lds_frags<HD>(qf, Qs + (qt * 16 + lr) * LDR, lane);
                lds_frags<HD>(dof, dOs + (qt * 16 + lr) * LDR, lane);
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    sacc = mfma16(qf[s], kf[s], sacc);     // S[query (lg*4+r)][key lr], log2 units
                    pacc = mfma16(dof[s], vf[s], pacc);    // dP same layout
                }
                const float4 L4 = *reinterpret_cast<const float4*>(L_s + qn0);
                const float4 D4 = *reinterpret_cast<const float4*>(D_s + qn0);
                const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // pad queries carry L = -inf (P = 0); pad-key lanes compute finite garbage in output columns
                    // that are never stored, so neither needs a select here
                    const float pr = __builtin_amdgcn_exp2f(sacc[r] + Lr[r]);
                    float ks = 1.f;
                    if (DROP)
                        ks = keep_scale(sd, (unsigned)((grp * G.g.nH + h) * N + s0 + qn0 + r), (unsigned)nk, G.drop_thresh,
                                        G.inv_keep);
                    pv[r] = pr * ks;
                    dsv[r] = pr * (pacc[r] * ks - Dr[r]);
                }
                pf.u[half * 2 + 0] = pack2bf(pv[0], pv[1]);
                pf.u[half * 2 + 1] = pack2bf(pv[2], pv[3]);
                dsf.u[half * 2 + 0] = pack2bf(dsv[0], dsv[1]);
                dsf.u[half * 2 + 1] = pack2bf(dsv[2], dsv[3]);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                Frag8 a;   // A[hd c*16+lr][kappa] = dO[query(kappa)][hd]
                const bool tail = (NKT & 1) && 2 * qp + 1 >= NKT;       // no second tile staged: a zero operand, not LDS beyond it
                a.u2[0] = tr4(dOs, LDR, (2 * qp) * 16 + lg * 4, c * 16, lr);
                a.u2[1] = tail ? make_uint2(0u, 0u) : tr4(dOs, LDR, (2 * qp + 1) * 16 + lg * 4, c * 16, lr);
                dvacc[c] = mfma16(a, pf, dvacc[c]);        // dVᵀ[hd (lg*4+r)][key lr]
                a.u2[0] = tr4(Qs, LDR, (2 * qp) * 16 + lg * 4, c * 16, lr);
                a.u2[1] = tail ? make_uint2(0u, 0u) : tr4(Qs, LDR, (2 * qp + 1) * 16 + lg * 4, c * 16, lr);
                dkacc[c] = mfma16(a, dsf, dkacc[c]);
            }
        }
        if (kv) {
            bf16_t* dvrow = dv + sp * G.dv_ps + krow * G.lddv + h * HD + lg * 4;    // several query parts: partial dk / dv each
            bf16_t* dkrow = dk + sp * G.dk_ps + krow * G.lddk + h * HD + lg * 4;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uint2 w;
                w.x = pack2bf(dvacc[c][0], dvacc[c][1]);
                w.y = pack2bf(dvacc[c][2], dvacc[c][3]);
                *reinterpret_cast<uint2*>(dvrow + c * 16) = w;
                w.x = pack2bf(dkacc[c][0] * (1.0f / LOG2E), dkacc[c][1] * (1.0f / LOG2E));
                w.y = pack2bf(dkacc[c][2] * (1.0f / LOG2E), dkacc[c][3] * (1.0f / LOG2E));
                *reinterpret_cast<uint2*>(dkrow + c * 16) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------- backward, ONE kernel: dQ, dK, dV (+ dS scratch)
// Window mode (round 4).  The two kernels above each recompute S, P, dP and dS.  Here that runs once, in a wave-specialised
// workgroup over one (window, head) whose Q' (= Q * scale * log2e), dO, -L and D = rowsum(dO . O) are staged in LDS up front:
//   compute waves (one per PAIR of key tiles) keep their k / v fragments and dK / dV accumulators in registers and walk the
//     queries in chunks of 32 (two query tiles).  Per chunk and key tile: S, dP (MFMA), P, dS (VALU), dV += dO^T P,
//     dK += Q'^T dS (MFMA) — and the bf16 dS tile goes to LDS as T[key][query] (double-buffered per chunk).
//   the service wave (the last one) turns the T tiles of the PREVIOUS chunk into dQ: the transposing LDS read returns dS keyed
//     the way the contraction over keys wants it (and exactly as the dS scratch of the table gradient stores it), K^T of all
//     key tiles sits in its registers, so dQ of a chunk is complete in one wave — no partial sums, no second pass over the
//     scores.  One raw barrier per chunk (LDS traffic only: nobody waits for the service wave's global stores).
// (A first version staged the chunks inside the loop, one step ahead, from the service wave: every step then waited for a
// global round trip — 223 us per stage-0 launch against 239 us for the two kernels it replaces.)
#ifndef ONE_MINW
#define ONE_MINW 4
#endif
#ifndef ONE_ABL            // ablation builds (tools/probes/attn_one_abl.sh): 1 no dS scratch stores, 2 no dQ work, 4 no T writes,
#define ONE_ABL 0          // 8 no dV / dK MFMAs, 16 no bias gathers, 32 no exp2
#endif
constexpr int ONE_CWAVES(int nkt) { return (nkt + 1) / 2; }   // compute waves: two key tiles each
// 196-token windows: one service wave with K^T in its registers, two 8-wave workgroups per CU.  392-token windows (25 key
// tiles, 13 compute waves): K^T of 13 tile pairs does not fit a wave's registers next to the dS fragments — it stays in LDS
// (rows unpadded to make room) — and TWO service waves take one query tile of the chunk each; one 15-wave workgroup per CU.
constexpr int ONE_SWAVES(int nkt) { return nkt > 16 ? 2 : 1; }
constexpr int ONE_LDR(int hd, int nkt) { return nkt > 16 ? hd : hd + 8; }
template <int HD, int NKT>
size_t one_lds(int tls) {
    return 2 * (size_t)(NKT * 16) * ONE_LDR(HD, NKT) * 2 + 5 * (size_t)NKT * 16 * 4 + (size_t)tls * 4 + 16 +
           2 * (size_t)ONE_CWAVES(NKT) * 4 * 256 * 2 + (NKT > 16 ? (size_t)NKT * 16 * HD * 2 : 0);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int HD, int NKT>
__global__ void __launch_bounds__((ONE_CWAVES(NKT) + ONE_SWAVES(NKT)) * 64, ONE_MINW) attn_bwd_one_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
    const bf16_t* __restrict__ o, const bf16_t* __restrict__ dout, const float* __restrict__ lse,
    const float* __restrict__ bias, const int* __restrict__ rid, bf16_t* __restrict__ dq, bf16_t* __restrict__ dk,
    bf16_t* __restrict__ dv, bf16_t* __restrict__ ds_out, float* __restrict__ trace, Geom G) {
    constexpr int NK = NKT * 16, KS = (HD + 31) / 32, LDR = ONE_LDR(HD, NKT), NC = HD / 16, CH = HD / 8;
    constexpr int NWC = ONE_CWAVES(NKT), NS = ONE_SWAVES(NKT), NTHR = (NWC + NS) * 64, NQP = (NKT + 1) / 2;
    constexpr bool KREG = NKT <= 16;                         // K^T in the service wave's registers (else read from LDS per use)
#ifdef ONE_TRACE           // probe build: s_memtime deltas of wave 0 (and the service wave's loop) into the unused dsum array
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#define ONE_T(i) if (trace && (threadIdx.x & 63) == 0) trace[(int64_t)blockIdx.x * 16 + (i)] = (float)(__builtin_amdgcn_s_memtime() - t_start)
#define ONE_TV(i, v) if (trace && (threadIdx.x & 63) == 0) trace[(int64_t)blockIdx.x * 16 + (i)] = (float)(v)
    unsigned long long bw_s = 0, bw_c = 0;       // cycles spent waiting at the per-chunk barrier (service / compute wave)
#define ONE_BAR(acc) { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); lds_barrier(); acc += __builtin_amdgcn_s_memtime() - t0_; }
#else
#define ONE_T(i)
#define ONE_TV(i, v)
#define ONE_BAR(acc) lds_barrier()
#endif
    constexpr int TW = 4 * 256;                              // bf16 of one compute wave's T tiles: [half][ti][16][16]
    static_assert(KS == 1 && CH == 4, "one-kernel backward: HD = 32");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Qs = reinterpret_cast<bf16_t*>(smem);            // [NK][LDR]  Q * scale * log2e
    bf16_t* dOs = Qs + NK * LDR;                             // [NK][LDR]
    float* L_s = reinterpret_cast<float*>(dOs + NK * LDR);   // [NK]  -lse * log2e (pad queries: -inf)
    float* D_s = L_s + NK;                                   // [NK]  rowsum(dO . O)
    int* row_s = reinterpret_cast<int*>(D_s + NK);           // [NK]
    int* linb_s = row_s + NK;
    int* rid_s = linb_s + NK;
    float* tab_s = reinterpret_cast<float*>(rid_s + NK);
    int* flag_s = reinterpret_cast<int*>(tab_s + G.tls);
    bf16_t* T_s = reinterpret_cast<bf16_t*>(flag_s + 4);     // [2][NWC][half][ti][16 keys][16 queries]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lg = lane >> 4, lr = lane & 15;
    const int gh = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = gh / G.g.nH + G.grp0, h = gh % G.g.nH;
    const int N = G.g.N;
    constexpr int nt = NKT;                                  // the launcher guarantees (N + 15) / 16 == NKT: no padding-only tile
    const bool tb = bias != nullptr;
    if (tid == 0) *flag_s = 0;
    token_rows<NK>(G, grp, row_s, tid, NTHR);                // ends with a barrier
    if (wave == 0) { ONE_T(0); }

    // ---- compute waves: the k / v fragments of their key tiles go out first (as in the dK / dV kernel)
    Frag8 kf[2], vf[2];
    float kmv[2];
    if (wave < NWC) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const int nk = (wave * 2 + ti) * 16 + lr;
            const bool kv = nk < N;
            const int64_t krow = row_s[kv ? nk : 0];
            Frag8 t1[KS], t2[KS];
            load_frags<HD>(t1, k + krow * G.g.ldk + h * HD, kv, lane);
            load_frags<HD>(t2, v + krow * G.g.ldv + h * HD, kv, lane);
            kf[ti] = t1[0];
            vf[ti] = t2[0];
            kmv[ti] = kv ? 0.f : -INFINITY;                  // pad keys: P = dS = 0 (they would enter dQ otherwise)
        }
    }
    // ---- staging, all threads: Q' and dO rows, D = rowsum(dO . O), -L; K rows (unpadded) into the second T buffer, which the
    // compute waves first write in step 1 — the service wave takes its K^T fragments from there right after the barrier
    static_assert(!KREG || (size_t)NK * HD <= (size_t)NWC * TW, "K rows fit one T buffer");
    bf16_t* Kst = T_s + (KREG ? 1 : 2) * NWC * TW;           // kept for the whole kernel: its own region behind the T buffers
    {
        // every global load of the prologue goes out before the first result is used: ONE round trip instead of five
        // (two staging passes, -L, the region ids, the bias table: 20 000 of a workgroup's 46 000 cycles)
        const float qmul = G.g.scale * LOG2E;
        constexpr int NP = (NK * CH + NTHR - 1) / NTHR;      // 16-byte pieces per thread and matrix
        constexpr int NTAB = 3;                              // table rows per thread: the band of a 4-frame window = 1 183 rows
        uint4 vq[NP], vd[NP], vo[NP], vk[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int idx = tid + j * NTHR, n = idx / CH, c = idx - n * CH;
            vq[j] = vd[j] = vo[j] = vk[j] = make_uint4(0, 0, 0, 0);
            if (idx < NK * CH && n < N) {
                const int64_t r = row_s[n];
                vq[j] = *reinterpret_cast<const uint4*>(q + r * G.g.ldq + h * HD + c * 8);
                vd[j] = *reinterpret_cast<const uint4*>(dout + r * G.g.ldo + h * HD + c * 8);
                vo[j] = *reinterpret_cast<const uint4*>(o + r * G.g.ldo + h * HD + c * 8);
                vk[j] = *reinterpret_cast<const uint4*>(k + r * G.g.ldk + h * HD + c * 8);
            }
        }
        static_assert(NK <= NTHR, "one token per thread");
        const int wloc = grp % G.nW;
        const float lv = (tid < N) ? lse[((int64_t)grp * G.g.nH + h) * N + tid] : 0.f;
        const int rv = (rid && tid < N) ? rid[wloc * N + tid] : 0, rv0 = rid ? rid[wloc * N] : 0;
        float tv[NTAB];
#pragma unroll
        for (int j = 0; j < NTAB; ++j) {
            const int i = tid + j * NTHR;
            tv[j] = i < G.tbn ? bias[(int64_t)(G.tb0 + i) * G.g.nH + h] : 0.f;      // [tlen][nH] parameter layout
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int idx = tid + j * NTHR, n = idx / CH, c = idx - n * CH;
            auto sc2 = [&](uint32_t w) { return pack2bf(half_lo(w) * qmul, half_hi(w) * qmul); };
            auto dot2 = [](uint32_t a, uint32_t b) {
                return half_lo(a) * half_lo(b) + half_hi(a) * half_hi(b);
            };
            float d = (dot2(vd[j].x, vo[j].x) + dot2(vd[j].y, vo[j].y)) + (dot2(vd[j].z, vo[j].z) + dot2(vd[j].w, vo[j].w));
            d += __shfl_xor(d, 1, 64);                       // the CH = 4 pieces of a row sit in adjacent lanes
            d += __shfl_xor(d, 2, 64);
            if (idx < NK * CH) {
                *reinterpret_cast<uint4*>(Qs + n * LDR + c * 8) = make_uint4(sc2(vq[j].x), sc2(vq[j].y), sc2(vq[j].z), sc2(vq[j].w));
                *reinterpret_cast<uint4*>(dOs + n * LDR + c * 8) = vd[j];
                *reinterpret_cast<uint4*>(Kst + n * HD + c * 8) = vk[j];
                if (c == 0) D_s[n] = d;
            }
        }
        if (tid < NK) {
            L_s[tid] = (tid < N) ? -lv * LOG2E : -INFINITY;
            linb_s[tid] = (tid < N) ? 4 * win_lin(G, tid) : 0;
            if (rid) {
                rid_s[tid] = rv;
                if (tid < N && rv != rv0) *flag_s = 1;
            }
        }
#pragma unroll
        for (int j = 0; j < NTAB; ++j) {
            const int i = tid + j * NTHR;
            if (i < G.tbn) tab_s[i] = tv[j] * LOG2E;
        }
    }
    if (wave == 0) { ONE_T(1); }
    lds_barrier();                                           // B0
    if (wave == 0) { ONE_T(2); }

    if (wave >= NWC) {
        // ================================================================= service wave(s)
        // K^T of every key-tile pair: A[hd c*16 + lr][kappa] = K[key(kappa)][hd], kappa = lg*8 + j: j < 4 -> tile 2 jp, key
        // lg*4 + j; j >= 4 -> tile 2 jp + 1.  Unscaled: dQ = (dS . K) * scale.
        auto kt_frag = [&](int jp, int c) {
            Frag8 f;
            f.u2[0] = tr4(Kst, HD, (2 * jp) * 16 + lg * 4, c * 16, lr);
            f.u2[1] = (2 * jp + 1 < NKT) ? tr4(Kst, HD, (2 * jp + 1) * 16 + lg * 4, c * 16, lr) : make_uint2(0u, 0u);
            return f;
        };
        Frag8 ktf[KREG ? NWC : 1][NC];
        if (KREG) {
#pragma unroll
            for (int jp = 0; jp < NWC; ++jp)
#pragma unroll
                for (int c = 0; c < NC; ++c) ktf[KREG ? jp : 0][c] = kt_frag(jp, c);
        }
        const int sw = wave - NWC;                           // two service waves: query tile `sw` of every chunk
        bf16_t* dsbase = (tb && ds_out) ? ds_out + (((int64_t)(grp - G.grp0) * G.g.nH + h) * nt * NKT * 64 + lane) * 4 : nullptr;
        // dQ of chunk pc from the T tiles in buffer bt.  All transposing reads of a query tile go out first, then the scratch
        // stores, then the MFMAs: with a read / wait / store per tile the wave spent 5 200 cycles per chunk and every compute
        // wave waited for it at the barrier.
        auto dq_chunk = [&](int pc, int bt) {
            const bf16_t* Tb = T_s + bt * (NWC * TW);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int qt = 2 * pc + half;
                if (qt >= NKT) break;
                if (NS == 2 && half != sw) continue;
                Frag8 bq[NWC];
#pragma unroll
                for (int jp = 0; jp < NWC; ++jp) {
                    bq[jp].u2[0] = tr4(Tb + jp * TW + (half * 2 + 0) * 256, 16, lg * 4, 0, lr);
                    bq[jp].u2[1] = (2 * jp + 1 < NKT) ? tr4(Tb + jp * TW + (half * 2 + 1) * 256, 16, lg * 4, 0, lr) : make_uint2(0u, 0u);
                }
                if (!(ONE_ABL & 1) && dsbase) {
                    bf16_t* dsq = dsbase + (int64_t)qt * NKT * 256;
#pragma unroll
                    for (int kt = 0; kt < NKT; ++kt) *reinterpret_cast<uint2*>(dsq + kt * 256) = bq[kt >> 1].u2[kt & 1];
                }
                f32x4_t qacc[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) qacc[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jp = 0; jp < NWC; ++jp)
#pragma unroll
                    for (int c = 0; c < NC; ++c)             // dQ^T[hd c*16+lg*4+r][query lr]
                        qacc[c] = mfma16(KREG ? ktf[KREG ? jp : 0][c] : kt_frag(jp, c), bq[jp], qacc[c]);
                const int qn = qt * 16 + lr;
                if (qn < N) {
                    bf16_t* drow = dq + (int64_t)row_s[qn] * G.lddq + h * HD + lg * 4;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        uint2 w2;
                        w2.x = pack2bf(qacc[c][0] * G.g.scale, qacc[c][1] * G.g.scale);
                        w2.y = pack2bf(qacc[c][2] * G.g.scale, qacc[c][3] * G.g.scale);
                        *reinterpret_cast<uint2*>(drow + c * 16) = w2;
                    }
                }
            }
        };
#pragma unroll 1
        for (int p = 0; p < NQP; ++p) {
            if (!(ONE_ABL & 2) && p > 0) dq_chunk(p - 1, (p - 1) & 1);
            ONE_BAR(bw_s);
        }
        ONE_T(5);
        ONE_TV(8, bw_s);
        if (!(ONE_ABL & 2)) dq_chunk(NQP - 1, (NQP - 1) & 1);
        ONE_T(6);
        return;
    }

    // ===================================================================== compute waves: key tiles 2 wave, 2 wave + 1
    int ko[2], rk[2];
    f32x4_t dvacc[2][NC], dkacc[2][NC];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            dvacc[ti][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            dkacc[ti][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
        const int nk = (wave * 2 + ti) * 16 + lr;
        rk[ti] = (rid && nk < N) ? rid_s[nk] : 0;
        ko[ti] = linb_s[nk < N ? nk : 0] - 4 * G.tcst;              // slot(q, key) = lin(q) - (lin(key) - tcst)
    }
    const bool masked = rid && *flag_s != 0;

    // One chunk of 32 queries (NH = 2) or the odd last tile (NH = 1), with / without the shift mask — four straight-line
    // bodies picked outside the loop: no branch inside, so the scheduler overlaps the LDS latencies of the four (query tile,
    // key tile) units of a chunk (a wave whose second key tile is padding computes it anyway: k = 0, P = 0).
    auto chunk = [&](int p, auto nh_c, auto masked_c) {
        constexpr int NH = decltype(nh_c)::value;
        constexpr bool MASKED = decltype(masked_c)::value;
        const bf16_t* Qb = Qs + p * 32 * LDR;
        const bf16_t* dOb = dOs + p * 32 * LDR;
        bf16_t* Tw = T_s + ((p & 1) * NWC + wave) * TW;
        Frag8 pf[2], dsf[2];
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) pf[ti].u4 = dsf[ti].u4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            const int qn0 = (2 * p + half) * 16 + lg * 4;    // this lane's 4 queries (window token ids)
            Frag8 qf, dof;
            qf.u4 = *reinterpret_cast<const uint4*>(Qb + (half * 16 + lr) * LDR + lg * 8);
            dof.u4 = *reinterpret_cast<const uint4*>(dOb + (half * 16 + lr) * LDR + lg * 8);
            const float4 L4 = *reinterpret_cast<const float4*>(L_s + qn0);
            const float4 D4 = *reinterpret_cast<const float4*>(D_s + qn0);
            const float Lr[4] = {L4.x, L4.y, L4.z, L4.w}, Dr[4] = {D4.x, D4.y, D4.z, D4.w};
            const int4 qb = *reinterpret_cast<const int4*>(linb_s + qn0);
            int4 rq4 = make_int4(0, 0, 0, 0);
            if (MASKED) rq4 = *reinterpret_cast<const int4*>(rid_s + qn0);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x4_t sacc, pacc = {0.f, 0.f, 0.f, 0.f};
                if (!(ONE_ABL & 16)) {
                    const char* tp = reinterpret_cast<const char*>(tab_s) - ko[ti];
                    sacc[0] = kmv[ti] + *reinterpret_cast<const float*>(tp + qb.x);
                    sacc[1] = kmv[ti] + *reinterpret_cast<const float*>(tp + qb.y);
                    sacc[2] = kmv[ti] + *reinterpret_cast<const float*>(tp + qb.z);
                    sacc[3] = kmv[ti] + *reinterpret_cast<const float*>(tp + qb.w);
                } else {
                    sacc = (f32x4_t){kmv[ti], kmv[ti], kmv[ti], kmv[ti]};
                }
                if (MASKED) {
                    sacc[0] += (rq4.x != rk[ti]) ? -100.0f * LOG2E : 0.0f;
                    sacc[1] += (rq4.y != rk[ti]) ? -100.0f * LOG2E : 0.0f;
                    sacc[2] += (rq4.z != rk[ti]) ? -100.0f * LOG2E : 0.0f;
                    sacc[3] += (rq4.w != rk[ti]) ? -100.0f * LOG2E : 0.0f;
                }
                sacc = mfma16(qf, kf[ti], sacc);             // S[query lg*4+r][key lr], log2 units
                pacc = mfma16(dof, vf[ti], pacc);            // dP, same layout
                float pv[4], dsv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[r] = (ONE_ABL & 32) ? sacc[r] + Lr[r] : __builtin_amdgcn_exp2f(sacc[r] + Lr[r]);     // pad keys / pad queries: exp2(-inf) = 0
                    dsv[r] = pv[r] * (pacc[r] - Dr[r]);
                }
                pf[ti].u[half * 2 + 0] = pack2bf(pv[0], pv[1]);
                pf[ti].u[half * 2 + 1] = pack2bf(pv[2], pv[3]);
                dsf[ti].u[half * 2 + 0] = pack2bf(dsv[0], dsv[1]);
                dsf[ti].u[half * 2 + 1] = pack2bf(dsv[2], dsv[3]);
                // dS tile as T[key lr][query lg*4 .. +3]: the service wave's transposing read returns it keyed the other way
                if (!(ONE_ABL & 4))
                *reinterpret_cast<uint2*>(Tw + (half * 2 + ti) * 256 + lr * 16 + lg * 4) =
                    make_uint2(dsf[ti].u[half * 2 + 0], dsf[ti].u[half * 2 + 1]);
            }
        }
#pragma unroll
        for (int c = 0; c < ((ONE_ABL & 8) ? 0 : NC); ++c) {
            Frag8 ad, aq;   // A[hd c*16+lr][kappa] = dO (Q')[query(kappa)][hd]
            ad.u2[0] = tr4(dOb, LDR, lg * 4, c * 16, lr);
            ad.u2[1] = NH == 2 ? tr4(dOb, LDR, 16 + lg * 4, c * 16, lr) : make_uint2(0u, 0u);   // odd tile count: nothing staged there
            aq.u2[0] = tr4(Qb, LDR, lg * 4, c * 16, lr);
            aq.u2[1] = NH == 2 ? tr4(Qb, LDR, 16 + lg * 4, c * 16, lr) : make_uint2(0u, 0u);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                dvacc[ti][c] = mfma16(ad, pf[ti], dvacc[ti][c]);      // dV^T[hd (lg*4+r)][key lr]
                dkacc[ti][c] = mfma16(aq, dsf[ti], dkacc[ti][c]);
            }
        }
    };
    auto loop = [&](auto masked_c) {
#pragma unroll 1
        for (int p = 0; p < NKT / 2; ++p) {
            chunk(p, std::integral_constant<int, 2>{}, masked_c);
            ONE_BAR(bw_c);
        }
        if (NKT & 1) {
            chunk(NKT / 2, std::integral_constant<int, 1>{}, masked_c);
            ONE_BAR(bw_c);
        }
    };
    if (masked) loop(std::true_type{});
    else loop(std::false_type{});
    if (wave == 0) { ONE_T(3); ONE_TV(7, bw_c); }
    if (wave == NWC - 1) { ONE_TV(9, bw_c); }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int nk = (wave * 2 + ti) * 16 + lr;
        if (nk < N) {
            const int64_t krow = row_s[nk];
            bf16_t* dvrow = dv + krow * G.lddv + h * HD + lg * 4;
            bf16_t* dkrow = dk + krow * G.lddk + h * HD + lg * 4;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                uint2 w;
                w.x = pack2bf(dvacc[ti][c][0], dvacc[ti][c][1]);
                w.y = pack2bf(dvacc[ti][c][2], dvacc[ti][c][3]);
                *reinterpret_cast<uint2*>(dvrow + c * 16) = w;
                w.x = pack2bf(dkacc[ti][c][0] * (1.0f / LOG2E), dkacc[ti][c][1] * (1.0f / LOG2E));
                w.y = pack2bf(dkacc[ti][c][2] * (1.0f / LOG2E), dkacc[ti][c][3] * (1.0f / LOG2E));
                *reinterpret_cast<uint2*>(dkrow + c * 16) = w;
            }
        }
    }
    if (wave == 0) { ONE_T(4); }
}

// d table[slot(q, key)][h] += sum over groups of dS[g][h][q][key], in two kernels and without atomics:
//   sum     partial[s][h][q tile][key tile][lane][4] (fp32) = sum over group slice s of the dQ kernel's bf16 scratch
//           (same fragment order; one thread = 4 scores; streaming, HBM-bound)
//   gather  one WAVE per (head, table row): its lanes share the <= N (query, key) pairs that map to the row
//           (query coords = key coords + the row's offset), add the slices' partial sums, wave-reduce, one writer per row
__global__ void __launch_bounds__(256) dbias_sum_kernel(const bf16_t* __restrict__ ds, float4* __restrict__ partial,
                                                        int groups, int64_t E2) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over nH * nqt * nkt * 32: 8 scores each
    if (e >= E2) return;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    // four groups per trip: their 16-byte loads are in flight together (a slice is long since round 5: few slices keep the
    // gather that follows short — it reads every slice's partial sum of every (table row, key) pair)
    int g = blockIdx.y;
    const int gs = gridDim.y;
    for (; g + 3 * gs < groups; g += 4 * gs) {
        Frag8 v0, v1, v2, v3;
        v0.u4 = *reinterpret_cast<const uint4*>(ds + ((int64_t)g * E2 + e) * 8);
        v1.u4 = *reinterpret_cast<const uint4*>(ds + ((int64_t)(g + gs) * E2 + e) * 8);
        v2.u4 = *reinterpret_cast<const uint4*>(ds + ((int64_t)(g + 2 * gs) * E2 + e) * 8);
        v3.u4 = *reinterpret_cast<const uint4*>(ds + ((int64_t)(g + 3 * gs) * E2 + e) * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (bf2f(v0.h[i]) + bf2f(v1.h[i])) + (bf2f(v2.h[i]) + bf2f(v3.h[i]));
    }
    for (; g < groups; g += gs) {
        Frag8 v;
        v.u4 = *reinterpret_cast<const uint4*>(ds + ((int64_t)g * E2 + e) * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += bf2f(v.h[i]);
    }
    float4* out = partial + ((int64_t)blockIdx.y * E2 + e) * 2;
    out[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    out[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

__device__ __forceinline__ void dbias_gather_wave(const float* __restrict__ dense, float* __restrict__ dtable, int nkt,
                                                  const Geom& G, int nsplit, int64_t split_stride, int w, int lane) {
    if (w >= G.tlen * G.g.nH) return;
    const int slot = w / G.g.nH, h = w - slot * G.g.nH;
    // slot = (dz + bwd-1) * ts_d + (dy + bwh-1) * ts_h + (dx + bww-1):  query coords = key coords + (dz, dy, dx)
    const int dz = slot / G.ts_d - (G.g.bwd - 1), rem = slot % G.ts_d;
    const int dy = rem / G.ts_h - (G.g.bwh - 1), dx = rem % G.ts_h - (G.g.bww - 1);
    const int nqt = (G.g.N + 15) >> 4, hw = G.g.bwh * G.g.bww;
    const float* dh = dense + (int64_t)h * nqt * nkt * 256;
    float a = 0.f;
    for (int key = lane; key < G.g.N; key += 64) {
        const int kz = key / hw, kr = key - kz * hw, ky = kr / G.g.bww, kx = kr - ky * G.g.bww;
        const int qz = kz + dz, qy = ky + dy, qx = kx + dx;
        if (qz < 0 || qz >= G.g.bwd || qy < 0 || qy >= G.g.bwh || qx < 0 || qx >= G.g.bww) continue;
        const int qn = (qz * G.g.bwh + qy) * G.g.bww + qx;
        if (qn >= G.g.N) continue;
        // fragment address of (qn, key): q tile qn>>4, key tile key>>4, lane = (key&15)>>2 << 4 | qn&15, r = key&3
        const float* pe = dh + (((int64_t)(qn >> 4) * nkt + (key >> 4)) * 64 + ((((key & 15) >> 2) << 4) | (qn & 15))) * 4 + (key & 3);
        for (int sp = 0; sp < nsplit; ++sp) a += pe[sp * split_stride];      // the slices' partial sums (was a fold launch)
    }
    a = wave_sum(a);
    if (lane == 0) dtable[w] += a;
}

__global__ void __launch_bounds__(256) dbias_gather_kernel(const float* __restrict__ dense, float* __restrict__ dtable,
                                                           int nkt, Geom G, int nsplit, int64_t split_stride) {
    dbias_gather_wave(dense, dtable, nkt, G, nsplit, split_stride, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// Table-driven gather: the fragment offset of every (table row, key) pair is a pure function of the window geometry, so
// it is built ONCE per geometry (dbias_index_kernel -> int32 [tlen][NK], -1 = no pair) and the per-step gather is loads
// only (the arithmetic version above spends ~2 integer divisions per key and wave: 21-43 us per launch).
__global__ void __launch_bounds__(256) dbias_index_kernel(int* __restrict__ tab, int nkt, Geom G) {
    const int NK = nkt * 16;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= G.tlen * NK) return;
    const int slot = e / NK, key = e - slot * NK;
    int off = -1;
    if (key < G.g.N) {
        const int dz = slot / G.ts_d - (G.g.bwd - 1), rem = slot % G.ts_d;
        const int dy = rem / G.ts_h - (G.g.bwh - 1), dx = rem % G.ts_h - (G.g.bww - 1);
        const int hw = G.g.bwh * G.g.bww;
        const int kz = key / hw, kr = key - kz * hw, ky = kr / G.g.bww, kx = kr - ky * G.g.bww;
        const int qz = kz + dz, qy = ky + dy, qx = kx + dx;
        const int qn = (qz * G.g.bwh + qy) * G.g.bww + qx;
        if (qz >= 0 && qz < G.g.bwd && qy >= 0 && qy < G.g.bwh && qx >= 0 && qx < G.g.bww && qn < G.g.N)
            off = (((qn >> 4) * nkt + (key >> 4)) * 64 + ((((key & 15) >> 2) << 4) | (qn & 15))) * 4 + (key & 3);
    }
    tab[e] = off;
}

__global__ void __launch_bounds__(256) dbias_gather_tab_kernel(const float* __restrict__ dense, float* __restrict__ dtable,
                                                               const int* __restrict__ tab, int nkt, Geom G, int nsplit,
                                                               int64_t split_stride, int slot0, int nslots) {
    // rows [slot0, slot0 + nslots): the band of the table a window of depth wd <= bwd can reach at all (8-frame clips
    // use 7 of the 15 temporal offsets of the (8, 7, 7) table) — the other rows have no (query, key) pair
    const int wl = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wl >= nslots * G.g.nH) return;
    const int slot = slot0 + wl / G.g.nH, h = wl % G.g.nH;
    const int w = slot * G.g.nH + h;
    const int NK = nkt * 16, nqt = (G.g.N + 15) >> 4;
    const float* dh = dense + (int64_t)h * nqt * nkt * 256;
    const int* trow = tab + slot * NK;
    float a = 0.f;
    for (int kb = lane; kb < NK; kb += 64) {
        const int off = trow[kb];
        if (off < 0) continue;
        const float* pe = dh + off;
        int sp = 0;
        for (; sp + 4 <= nsplit; sp += 4)
            a += (pe[sp * split_stride] + pe[(sp + 1) * split_stride]) + (pe[(sp + 2) * split_stride] + pe[(sp + 3) * split_stride]);
        for (; sp < nsplit; ++sp) a += pe[sp * split_stride];
    }
    a = wave_sum(a);
    if (lane == 0) dtable[w] += a;
}

// The gathers of SEVERAL attention blocks as one launch (clv_attn_dbias_gather_batch): nothing reads a table gradient
// before the optimizer, so a backward segment leaves the slices' partial sums in its work buffers (stage bit 8 of
// clv_attn_bwd) and gathers all blocks at its end — 12 latency-bound launches per step become one.
struct GatherTable {
    ClvDbiasGather e[CLV_DBIAS_GATHER_MAX];
    int n;
};
// Before the gather: the slices' partial sums of every entry are added into slice 0 by a streaming pass (float4, coalesced).
// The gather reads 4-byte elements along the table rows' diagonals — every one a sector fetch — so it is run on ONE slice
// (326 MB of fetches for 46 MB of partial sums otherwise: 104 us at the end of the step's backward).
__global__ void __launch_bounds__(256) dbias_split_sum_batch_kernel(GatherTable tab) {
    int idx = 0;
    for (int i = 1; i < tab.n; ++i)
        if ((int)blockIdx.x >= tab.e[i].block_begin) idx = i;
    const ClvDbiasGather& en = tab.e[idx];
    const int64_t n4 = en.split_stride >> 2;                   // float4 elements per slice (the stride is a multiple of 256)
    const int64_t i4 = ((int64_t)((int)blockIdx.x - en.block_begin)) * 256 + threadIdx.x;
    if (i4 >= n4 || en.nsplit <= 1) return;
    float4* base = static_cast<float4*>(const_cast<void*>(en.partial));
    float4 a = base[i4];
    for (int sp = 1; sp < en.nsplit; ++sp) {
        const float4 b = base[i4 + sp * n4];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    base[i4] = a;
}

__global__ void __launch_bounds__(256) dbias_gather_batch_kernel(GatherTable tab) {
    int idx = 0;
    for (int i = 1; i < tab.n; ++i)
        if ((int)blockIdx.x >= tab.e[i].block_begin) idx = i;
    const ClvDbiasGather& en = tab.e[idx];
    // one wave = one table row x a group of hg <= 4 heads: the row's index entries are loaded once and the heads' elements
    // (the same offsets in each head's block) are in flight together
    const int hg = en.pad > 0 ? en.pad : 1, ngrp = en.nH / hg;
    const int wl = ((int)blockIdx.x - en.block_begin) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wl >= en.nslots * ngrp) return;
    const int slot = en.slot0 + wl / ngrp, h0 = (wl % ngrp) * hg;
    const int NK = en.nkt * 16, nqt = (en.N + 15) >> 4;
    const int64_t hstride = (int64_t)nqt * en.nkt * 256;
    const float* dh = static_cast<const float*>(en.partial) + h0 * hstride;
    const int* trow = static_cast<const int*>(en.index) + slot * NK;
    const int nsplit = en.nsplit;
    const int64_t split_stride = en.split_stride;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    // four keys per trip: the four index loads, then the (dependent, scattered) element loads, are in flight together — the
    // kernel is a chain of two dependent loads per key and nothing else
    for (int kb0 = lane; kb0 < NK; kb0 += 256) {
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) off[u] = kb0 + 64 * u < NK ? trow[kb0 + 64 * u] : -1;
        if (nsplit == 1) {
#pragma unroll
            for (int hh = 0; hh < 4; ++hh) {
                if (hh >= hg) break;
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = off[u] >= 0 ? dh[hh * hstride + off[u]] : 0.f;
                a[hh] += (v[0] + v[1]) + (v[2] + v[3]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (off[u] < 0) continue;
                const float* pe = dh + off[u];
                int sp = 0;
                for (; sp + 4 <= nsplit; sp += 4)
                    a[0] += (pe[sp * split_stride] + pe[(sp + 1) * split_stride]) +
                            (pe[(sp + 2) * split_stride] + pe[(sp + 3) * split_stride]);
                for (; sp < nsplit; ++sp) a[0] += pe[sp * split_stride];
            }
        }
    }
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
        if (hh >= hg) break;
        const float t = wave_sum(a[hh]);
        if (lane == 0) static_cast<float*>(en.dtable)[slot * en.nH + h0 + hh] += t;
    }
}

// ------------------------------------------------------------------------- fp32 parity kernel
// The "parity mode" forward (clv_attn_f32_fwd): q / k / v / o in fp32 storage, every product and the softmax in fp32
// on the VALU — through the SAME index logic as the bf16 MFMA kernels above (tok_row: roll / partition / reverse;
// win_lin + tcst: the relative-position table row; region-id compare for the shift mask; additive key mask).  It exists
// so that the step can be evaluated with fp32 arithmetic end to end and compared with the reference's CPU fp32 path at
// 1e-3 on the losses (tests/test_parity_gpu.py); it is not on the training path.  One workgroup = one (group, head)
// and a slice of its queries; one wave = one query row at a time: lane l holds the scores of keys l, l + 64, ...
constexpr int F32_MAXT = 16;            // <= 1024 keys
__global__ void __launch_bounds__(256) attn_f32_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ o,
                                                           const float* __restrict__ table, const int* __restrict__ rid,
                                                           const float* __restrict__ kmask, Geom G, int round_p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = G.g.N, hd = G.g.hd;
    int* row_s = reinterpret_cast<int*>(smem);                  // token -> tensor row
    int* lin_s = row_s + N;                                     // lin(n) of the table's window
    int* rid_s = lin_s + N;                                     // region id
    float* kadd_s = reinterpret_cast<float*>(rid_s + N);        // additive key mask
    float* p_s = kadd_s + N;                                    // [4 waves][N] probabilities
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gh = blockIdx.x, grp = gh / G.g.nH, h = gh - grp * G.g.nH;
    const int wloc = (G.g.mode == 1) ? grp % G.nW : 0;
    for (int n = tid; n < N; n += 256) {
        row_s[n] = (int)tok_row(G, grp, n);
        lin_s[n] = (G.g.mode == 1 && table) ? win_lin(G, n) : 0;
        rid_s[n] = (G.g.mode == 1 && rid) ? rid[wloc * N + n] : 0;
        kadd_s[n] = (G.g.mode == 0 && kmask) ? kmask[(int64_t)grp * N + n] : 0.f;
    }
    __syncthreads();
    float* pw = p_s + wave * N;
    for (int qi = blockIdx.y * 4 + wave; qi < N; qi += 4 * gridDim.y) {
        const float* qp = q + (int64_t)row_s[qi] * G.g.ldq + h * hd;
        const int linq = lin_s[qi] + G.tcst + G.tb0, rq = rid_s[qi];
        float s[F32_MAXT];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < F32_MAXT; ++t) {
            const int j = t * 64 + lane;
            s[t] = -INFINITY;
            if (j < N) {
                const float* kp = k + (int64_t)row_s[j] * G.g.ldk + h * hd;
                float dot = 0.f;
                for (int d = 0; d < hd; d += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(qp + d);
                    const float4 b = *reinterpret_cast<const float4*>(kp + d);
                    dot = fmaf(a.x, b.x, dot); dot = fmaf(a.y, b.y, dot);
                    dot = fmaf(a.z, b.z, dot); dot = fmaf(a.w, b.w, dot);
                }
                float sc = dot * G.g.scale;
                if (G.g.mode == 1 && table) sc += table[(int64_t)(linq - lin_s[j]) * G.g.nH + h];
                if (G.g.mode == 1 && rid && rid_s[j] != rq) sc += -100.0f;
                sc += kadd_s[j];
                s[t] = sc;
                m = fmaxf(m, sc);
            }
        }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < F32_MAXT; ++t) {
            const int j = t * 64 + lane;
            if (j < N) {
                float e = expf(s[t] - m);
                if (round_p == 1) e = bf2f(f2bf(e));  // emulate the bf16 P operand of the MFMA kernels
                else if (round_p == 2) e = (float)(_Float16)e;     // ... or an f16 one (the f16-forward study)
                s[t] = e;
                sum += e;
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int t = 0; t < F32_MAXT; ++t) {
            const int j = t * 64 + lane;
            if (j < N) pw[j] = s[t];
        }
        __builtin_amdgcn_wave_barrier();                  // a wave's own LDS operations execute in order
        if (lane < hd) {
            float acc = 0.f;
            for (int j = 0; j < N; ++j) acc = fmaf(pw[j], v[(int64_t)row_s[j] * G.g.ldv + h * hd + lane], acc);
            o[(int64_t)row_s[qi] * G.g.ldo + h * hd + lane] = acc * inv;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// fp32 backward of the same arithmetic (parity mode: gradients of the step checked against the reference's at fp32
// tolerances, tests/test_parity_gpu.py).  Two passes over the score matrix, both recomputing it through the forward's index
// helpers, so that every output element has ONE writer (only the relative-position table is accumulated with atomics):
//   pass Q (one wave per query row i):  lse_i, delta_i = sum_j p_ij dp_ij with dp_ij = dO_i . v_j;  dS_ij = p_ij (dp_ij - delta_i);
//                                       dq_i = scale sum_j dS_ij k_j;  d table[row(i, j)] += dS_ij
//   pass K (one wave per key row j):    p_ij = exp(s_ij - lse_i), dS_ij as above from the stored lse / delta;
//                                       dk_j = scale sum_i dS_ij q_i;  dv_j = sum_i p_ij dO_i
template <bool PASS_K>
__global__ void __launch_bounds__(256) attn_f32_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ dout,
                                                           const float* __restrict__ table, const int* __restrict__ rid,
                                                           const float* __restrict__ kmask, float* __restrict__ dq,
                                                           float* __restrict__ dk, float* __restrict__ dv,
                                                           float* __restrict__ dtable, float* __restrict__ lse_w,
                                                           float* __restrict__ delta_w, Geom G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int N = G.g.N, hd = G.g.hd;
    int* row_s = reinterpret_cast<int*>(smem);
    int* lin_s = row_s + N;
    int* rid_s = lin_s + N;
    float* kadd_s = reinterpret_cast<float*>(rid_s + N);
    float* p_s = kadd_s + N;                                    // [4 waves][2][N]: dS (and P in pass K)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gh = blockIdx.x, grp = gh / G.g.nH, h = gh - grp * G.g.nH;
    const int wloc = (G.g.mode == 1) ? grp % G.nW : 0;
    for (int n = tid; n < N; n += 256) {
        row_s[n] = (int)tok_row(G, grp, n);
        lin_s[n] = (G.g.mode == 1 && table) ? win_lin(G, n) : 0;
        rid_s[n] = (G.g.mode == 1 && rid) ? rid[wloc * N + n] : 0;
        kadd_s[n] = (G.g.mode == 0 && kmask) ? kmask[(int64_t)grp * N + n] : 0.f;
    }
    __syncthreads();
    float* ds_w = p_s + wave * 2 * N;
    float* pp_w = ds_w + N;
    const int64_t stat0 = (int64_t)gh * N;
    auto score = [&](int i, int j) {                              // s_ij exactly as the forward builds it
        const float* qp = q + (int64_t)row_s[i] * G.g.ldq + h * hd;
        const float* kp = k + (int64_t)row_s[j] * G.g.ldk + h * hd;
        float dot = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const float4 a = *reinterpret_cast<const float4*>(qp + d);
            const float4 b = *reinterpret_cast<const float4*>(kp + d);
            dot = fmaf(a.x, b.x, dot); dot = fmaf(a.y, b.y, dot);
            dot = fmaf(a.z, b.z, dot); dot = fmaf(a.w, b.w, dot);
        }
        float sc = dot * G.g.scale;
        if (G.g.mode == 1 && table) sc += table[(int64_t)(lin_s[i] + G.tcst + G.tb0 - lin_s[j]) * G.g.nH + h];
        if (G.g.mode == 1 && rid && rid_s[j] != rid_s[i]) sc += -100.0f;
        return sc + kadd_s[j];
    };
    auto dotrow = [&](const float* a, const float* b) {
        float dot = 0.f;
        for (int d = 0; d < hd; d += 4) {
            const float4 x = *reinterpret_cast<const float4*>(a + d);
            const float4 y = *reinterpret_cast<const float4*>(b + d);
            dot = fmaf(x.x, y.x, dot); dot = fmaf(x.y, y.y, dot);
            dot = fmaf(x.z, y.z, dot); dot = fmaf(x.w, y.w, dot);
        }
        return dot;
    };
    if (!PASS_K) {
        for (int qi = blockIdx.y * 4 + wave; qi < N; qi += 4 * gridDim.y) {
            const float* dop = dout + (int64_t)row_s[qi] * G.g.ldo + h * hd;
            float s[F32_MAXT], dp[F32_MAXT];
            float m = -INFINITY;
#pragma unroll
            for (int t = 0; t < F32_MAXT; ++t) {
                const int j = t * 64 + lane;
                s[t] = -INFINITY;
                dp[t] = 0.f;
                if (j < N) {
                    s[t] = score(qi, j);
                    dp[t] = dotrow(dop, v + (int64_t)row_s[j] * G.g.ldv + h * hd);
                    m = fmaxf(m, s[t]);
                }
            }
            m = wave_max(m);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < F32_MAXT; ++t)
                if (t * 64 + lane < N) { s[t] = expf(s[t] - m); sum += s[t]; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            float delta = 0.f;
#pragma unroll
            for (int t = 0; t < F32_MAXT; ++t)
                if (t * 64 + lane < N) { s[t] *= inv; delta = fmaf(s[t], dp[t], delta); }
            delta = wave_sum(delta);
            if (lane == 0) { lse_w[stat0 + qi] = m + logf(sum); delta_w[stat0 + qi] = delta; }
#pragma unroll
            for (int t = 0; t < F32_MAXT; ++t) {
                const int j = t * 64 + lane;
                if (j < N) {
                    const float ds = s[t] * (dp[t] - delta);
                    ds_w[j] = ds;
                    if (G.g.mode == 1 && table && dtable)
                        atomicAdd(dtable + (int64_t)(lin_s[qi] + G.tcst + G.tb0 - lin_s[j]) * G.g.nH + h, ds);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < hd) {
                float acc = 0.f;
                for (int j = 0; j < N; ++j) acc = fmaf(ds_w[j], k[(int64_t)row_s[j] * G.g.ldk + h * hd + lane], acc);
                dq[(int64_t)row_s[qi] * G.g.ldq + h * hd + lane] = acc * G.g.scale;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (int kj = blockIdx.y * 4 + wave; kj < N; kj += 4 * gridDim.y) {
            const float* vp = v + (int64_t)row_s[kj] * G.g.ldv + h * hd;
#pragma unroll
            for (int t = 0; t < F32_MAXT; ++t) {
                const int i = t * 64 + lane;
                if (i < N) {
                    const float p = expf(score(i, kj) - lse_w[stat0 + i]);
                    const float dpv = dotrow(dout + (int64_t)row_s[i] * G.g.ldo + h * hd, vp);
                    pp_w[i] = p;
                    ds_w[i] = p * (dpv - delta_w[stat0 + i]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < hd) {
                float ak = 0.f, av = 0.f;
                for (int i = 0; i < N; ++i) {
                    ak = fmaf(ds_w[i], q[(int64_t)row_s[i] * G.g.ldq + h * hd + lane], ak);
                    av = fmaf(pp_w[i], dout[(int64_t)row_s[i] * G.g.ldo + h * hd + lane], av);
                }
                dk[(int64_t)row_s[kj] * G.g.ldk + h * hd + lane] = ak * G.g.scale;
                dv[(int64_t)row_s[kj] * G.g.ldv + h * hd + lane] = av;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------------- long sequences: merging the parts
// Forward: key part p delivered o_p = softmax over ITS keys . V_p and lse_p; over all keys
//   lse = log sum_p exp(lse_p),   o = sum_p exp(lse_p - lse) o_p.
// One thread = 8 consecutive channels of one (token, head).
__global__ void __launch_bounds__(256) seq_combine_fwd_kernel(const bf16_t* __restrict__ o_part, const float* __restrict__ lse_part,
                                                              bf16_t* __restrict__ o, float* __restrict__ lse, Geom G) {
    const int HD = G.g.hd, C = G.g.nH * HD, cpr = C / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tokens = (int64_t)G.g.groups * G.g.N;
    if (idx >= tokens * cpr) return;
    const int64_t row = idx / cpr;
    const int c = (int)(idx - row * cpr) * 8, h = c / HD;
    const int64_t grp = row / G.g.N, n = row - grp * G.g.N;
    const int64_t li = (grp * G.g.nH + h) * G.g.N + n;
    float lp[4], m = -INFINITY;
    for (int p = 0; p < G.nparts; ++p) {
        lp[p] = lse_part[p * G.lse_ps + li];
        m = fmaxf(m, lp[p]);
    }
    float tot = 0.f;
    for (int p = 0; p < G.nparts; ++p) tot += __expf(lp[p] - m);
    const float L = m + __logf(tot);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int p = 0; p < G.nparts; ++p) {
        const float w = __expf(lp[p] - L);
        Frag8 f;
        f.u4 = *reinterpret_cast<const uint4*>(o_part + p * G.o_ps + row * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(w, bf2f(f.h[e]), acc[e]);
    }
    *reinterpret_cast<uint4*>(o + row * G.g.ldo + c) =
        make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
    if ((c % HD) == 0) lse[li] = L;
}

// Backward: dq is the sum of the key parts' partial dq, dk / dv the sum of the query parts' partial dk / dv.
__global__ void __launch_bounds__(256) seq_combine_bwd_kernel(const bf16_t* __restrict__ part, bf16_t* __restrict__ dq,
                                                              bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, Geom G) {
    const int C = G.g.nH * G.g.hd, cpr = C / 8;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t tokens = (int64_t)G.g.groups * G.g.N;
    if (idx >= 3 * tokens * cpr) return;
    const int which = (int)(idx / (tokens * cpr));
    const int64_t rem = idx - which * tokens * cpr, row = rem / cpr;
    const int c = (int)(rem - row * cpr) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int p = 0; p < G.nparts; ++p) {
        Frag8 f;      // scratch layout [part][dq | dk | dv][tokens][C]
        f.u4 = *reinterpret_cast<const uint4*>(part + ((int64_t)p * 3 + which) * tokens * C + row * C + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += bf2f(f.h[e]);
    }
    bf16_t* out = which == 0 ? dq + row * G.g.ldq : which == 1 ? dk + row * G.g.ldk : dv + row * G.g.ldv;
    *reinterpret_cast<uint4*>(out + c) =
        make_uint4(pack2bf(acc[0], acc[1]), pack2bf(acc[2], acc[3]), pack2bf(acc[4], acc[5]), pack2bf(acc[6], acc[7]));
}

// ------------------------------------------------------------------------- host side
constexpr int SEQ_ONE_PART_TILES = 28;       // 448 keys: the largest instantiation (LDS)
bool make_geom(const ClvAttnGeom* g, Geom& G) {
    if (!g) return false;
    G.g = *g;
    G.nWh = G.nWw = G.nW = 1;
    G.drop_thresh = 0;
    G.inv_keep = 1.f;
    if (g->dropout_p < 0.f || g->dropout_p >= 1.f) return false;
    if (g->dropout_p > 0.f) {
        G.drop_thresh = (unsigned)((double)g->dropout_p * 4294967296.0);
        G.inv_keep = 1.0f / (1.0f - g->dropout_p);
    }
    if (g->N <= 0 || g->nH <= 0 || g->groups <= 0) return false;
    if (g->hd != 16 && g->hd != 32 && g->hd != 64) return false;
    if ((g->ldq | g->ldk | g->ldv | g->ldo) & 7) return false;   // 16-byte row alignment
    G.ts_d = G.ts_h = G.tcst = G.tlen = G.tls = G.tb0 = G.tbn = 0;
    {
        const int pairs = g->groups * g->nH, tiles = (g->N + 15) / 16;
        G.tsplit = 1;
        while (G.tsplit < 4 && pairs * G.tsplit < 384 && tiles >= 8 * G.tsplit) G.tsplit *= 2;
        G.nparts = 1;
        G.grp0 = 0;
        G.pt16 = tiles * 16;
        G.lddq = g->ldq; G.lddk = g->ldk; G.lddv = g->ldv;
        G.o_ps = G.lse_ps = G.dq_ps = G.dk_ps = G.dv_ps = 0;
        // K / V (Q / dO) of one (sample, head) exceed the LDS: two parts (beyond two parts' reach the bf16 launchers find
        // no instantiation and report CLV_ERR_UNSUPPORTED; the fp32 parity kernel stages nothing and takes any length)
        if (g->mode == 0 && tiles > SEQ_ONE_PART_TILES && tiles <= 2 * SEQ_ONE_PART_TILES) {
            G.nparts = 2;
            G.pt16 = (tiles + 1) / 2 * 16;
            G.tsplit = 2;                                   // looped tiles per workgroup set: <= 64 (see launch checks)
            const int64_t tokens = (int64_t)g->groups * g->N, C = (int64_t)g->nH * g->hd;
            G.o_ps = tokens * C;
            G.lse_ps = (int64_t)g->groups * g->nH * g->N;
            G.lddq = G.lddk = G.lddv = (int)C;
            G.dq_ps = G.dk_ps = G.dv_ps = 3 * tokens * C;   // scratch [part][dq | dk | dv][tokens][C]
        }
    }
    if (g->mode == 1) {
        if (g->wd <= 0 || g->wh <= 0 || g->ww <= 0) return false;
        if (g->D % g->wd || g->H % g->wh || g->W % g->ww) return false;
        if (g->N != g->wd * g->wh * g->ww) return false;
        G.nWh = g->H / g->wh;
        G.nWw = g->W / g->ww;
        G.nW = (g->D / g->wd) * G.nWh * G.nWw;
        if (g->groups % G.nW) return false;
        if (g->sd < 0 || g->sd >= g->D || g->sh < 0 || g->sh >= g->H || g->sw < 0 || g->sw >= g->W) return false;
        if (g->bwd || g->bwh || g->bww) {                    // relative-position table of window (bwd, bwh, bww)
            if (g->bwd <= 0 || g->bwh <= 0 || g->bww <= 0) return false;
            G.ts_h = 2 * g->bww - 1;
            G.ts_d = (2 * g->bwh - 1) * G.ts_h;
            G.tlen = (2 * g->bwd - 1) * G.ts_d;
            // temporal offsets |dz| <= wd - 1 only: 8-frame clips use 7 of the 15 of the (8, 7, 7) table — the kernels
            // keep just that band in LDS (4.7 instead of 10.1 KB, half the strided table loads per workgroup)
            G.tb0 = g->wd < g->bwd ? (g->bwd - g->wd) * G.ts_d : 0;
            G.tbn = g->wd < g->bwd ? (2 * g->wd - 1) * G.ts_d : G.tlen;
            G.tcst = (g->bwd - 1) * G.ts_d + (g->bwh - 1) * G.ts_h + (g->bww - 1) - G.tb0;
            G.tls = (G.tbn + 3) / 4 * 4;
            // every token id n < N, decomposed by the TABLE's window, must stay inside that window
            if (g->N > g->bwd * g->bwh * g->bww) return false;
        }
    } else if (g->mode != 0) {
        return false;
    }
    return true;
}

constexpr size_t MAX_LDS = 160 * 1024;
constexpr int DBIAS_SPLITS = 32;      // group slices of the dS reduction (partial sums, no atomics)

// bytes of the bf16 dS scratch (16 x 16 fragments of every (group, head)), rounded so that the fp32 dense sums that
// follow it in `work` stay 256-byte aligned
// Group slices of the streaming dS sum: 64 groups each (round 5; 16 before).  Every slice's partial sum of every (table row,
// key) pair is read by the gather, so few long slices (four loads in flight per thread in dbias_sum_kernel) beat many
// short ones: same-box 11.41 / 11.44 -> 11.29 / 11.34 ms per step.
inline int dbias_splits(const Geom& G) {
    static const int gps = getenv("CLV_DBIAS_GROUPS_PER_SLICE") ? atoi(getenv("CLV_DBIAS_GROUPS_PER_SLICE")) : 64;
    int splits = G.g.groups / (gps > 0 ? gps : 64);
    return splits < 1 ? 1 : (splits > DBIAS_SPLITS ? DBIAS_SPLITS : splits);
}
inline int64_t ds_scratch_bytes(const Geom& G, int nkt) {
    const int64_t b = (int64_t)G.g.groups * G.g.nH * ((G.g.N + 15) / 16) * nkt * 256 * 2;
    return (b + 255) / 256 * 256;
}

template <int HD, int NKT>
size_t fwd_lds(int tls, bool window = false) { return 2 * (size_t)(NKT * 16) * (HD + 8) * 2 + (window ? 3 : 4) * (size_t)NKT * 16 * 4 + (size_t)tls * 4 + 16; }
template <int HD, int NKT>
size_t dq_lds(int tls) { return 2 * (size_t)(NKT * 16) * (HD + 8) * 2 + 4 * (size_t)NKT * 16 * 4 + (size_t)tls * 4 + 16; }
template <int HD, int NKT>
size_t dkv_lds(int tls) { return 2 * (size_t)(NKT * 16) * (HD + 8) * 2 + 5 * (size_t)NKT * 16 * 4 + (size_t)tls * 4 + 16; }

// Three compiled variants per (HD, NKT): window (bias/rid, no dropout), sequence, sequence + dropout.
#define CLV_PICK_N(KERNEL, NT, ...)                                                                    \
    do {                                                                                               \
        if (G.g.mode == 1) { KERNEL<HD, NT, false, 1> __VA_ARGS__; }                                   \
        else if (G.drop_thresh) { KERNEL<HD, NT, true, 0> __VA_ARGS__; }                               \
        else { KERNEL<HD, NT, false, 0> __VA_ARGS__; }                                                 \
    } while (0)
#define CLV_PICK(KERNEL, ...) CLV_PICK_N(KERNEL, NKT, __VA_ARGS__)

// dynamic LDS above 64 KB must be opted into per kernel; the request varies with the table length, so opt in to
// the whole 160 KB once and pass the actual size at launch
template <int HD, int NKT>
void set_attrs() {
    static bool done = false;
    if (done) return;
    done = true;
#define CLV_ATTR(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, (int)MAX_LDS)
    CLV_ATTR((attn_fwd_kernel<HD, NKT, false, 1>));
    CLV_ATTR((attn_fwd_kernel<HD, NKT, false, 0>));
    CLV_ATTR((attn_fwd_kernel<HD, NKT, true, 0>));
    CLV_ATTR((attn_bwd_dq_kernel<HD, NKT, false, 1>));
    CLV_ATTR((attn_bwd_dq_kernel<HD, NKT, false, 0>));
    CLV_ATTR((attn_bwd_dq_kernel<HD, NKT, true, 0>));
    constexpr int NKE = (NKT + 1) & ~1;                       // dK / dV: even tile count (see launch_bwd)
    CLV_ATTR((attn_bwd_dkv_kernel<HD, NKE, false, 1>));
    CLV_ATTR((attn_bwd_dkv_kernel<HD, NKE, false, 0>));
    CLV_ATTR((attn_bwd_dkv_kernel<HD, NKE, true, 0>));
    if constexpr (HD == 32 && (NKT == 13 || NKT == 25)) CLV_ATTR((attn_bwd_one_kernel<HD, NKT>));
#undef CLV_ATTR
}

template <int HD, int NKT>
int launch_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const float* bias,
               const int32_t* rid, const float* kmask, const unsigned long long* seed, const Geom& G, hipStream_t st) {
    const int bl = bias ? G.tls : 0;
    const size_t lds = fwd_lds<HD, NKT>(bl, G.g.mode == 1);
    if (lds > MAX_LDS || dq_lds<HD, NKT>(bl) > MAX_LDS || dkv_lds<HD, NKT>(bl) > MAX_LDS) return CLV_ERR_UNSUPPORTED;
    if (G.g.mode == 1 && G.drop_thresh) return CLV_ERR_UNSUPPORTED;
    set_attrs<HD, NKT>();
    const int nblk = G.g.groups * G.g.nH * G.tsplit * G.nparts;
    if (G.nparts == 1) {
        CLV_PICK(attn_fwd_kernel, <<<dim3(nblk), dim3(THREADS), lds, st>>>((const bf16_t*)q, (const bf16_t*)k,
                 (const bf16_t*)v, (bf16_t*)o, lse, bias, rid, kmask, seed, G));
        return clv_check_launch();
    }
    // two key parts: partial o (dense [tokens][C]) + lse per part into the scratch, then the merge
    if (!G.g.work || (NKT + WAVES - 1) / WAVES * WAVES * G.tsplit < (G.g.N + 15) / 16) return CLV_ERR_UNSUPPORTED;
    Geom P = G;
    P.g.ldo = G.g.nH * G.g.hd;
    bf16_t* o_part = reinterpret_cast<bf16_t*>(G.g.work);
    float* lse_part = reinterpret_cast<float*>(o_part + G.nparts * G.o_ps);
    CLV_PICK(attn_fwd_kernel, <<<dim3(nblk), dim3(THREADS), lds, st>>>((const bf16_t*)q, (const bf16_t*)k,
             (const bf16_t*)v, o_part, lse_part, bias, rid, kmask, seed, P));
    int rc = clv_check_launch();
    if (rc) return rc;
    const int64_t items = (int64_t)G.g.groups * G.g.N * (G.g.nH * G.g.hd / 8);
    hipLaunchKernelGGL(seq_combine_fwd_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, o_part, lse_part,
                       (bf16_t*)o, lse, G);
    return clv_check_launch();
}

// Window mode, head dim 32, the two Swin window sizes: ONE kernel for dQ / dK / dV (+ the dS scratch), see
// attn_bwd_one_kernel.  CLV_ATTN_BWD_ONE: 0 = the two-kernel path, 1 (default) = when there are (group, head) pairs for
// most of the chip (>= 192: Swin-B stage 3 at 8 clips, 256 pairs, 143 -> 101 us), 2 = whenever the shapes allow (tests).
template <int HD, int NKT>
bool one_eligible(const Geom& G, bool has_bias) {
    if constexpr (HD == 32 && (NKT == 13 || NKT == 25)) {
        const char* one_env = getenv("CLV_ATTN_BWD_ONE");     // read per call: the tests switch it
        const int one_mode = one_env ? atoi(one_env) : 1;
        return one_mode > 0 && G.g.mode == 1 && G.nparts == 1 && !G.drop_thresh &&
               (G.tsplit == 1 || G.g.groups * G.g.nH >= 192 || one_mode > 1) && (G.g.N + 15) / 16 == NKT && has_bias &&
               G.tbn <= 3 * (ONE_CWAVES(NKT) + ONE_SWAVES(NKT)) * 64 && one_lds<HD, NKT>(G.tls) <= MAX_LDS;
    }
    return false;
}
template <int HD, int NKT>
int query_one(const Geom& G) { return one_eligible<HD, NKT>(G, true) ? 1 : 0; }

template <int HD, int NKT>
int launch_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
               const float* lse, const float* bias, const int32_t* rid, const float* kmask, void* dq,
               void* dk, void* dv, float* dbias, float* dsum, void* work, const unsigned long long* seed, int stages,
               const Geom& G, hipStream_t st) {
    const int bl = bias ? G.tls : 0;
    // the dK / dV kernel walks query tiles in pairs inside a rolled loop: an odd count costs it more (zeroed half
    // operands, a break) than the one padding tile of the next even count (392 tokens: 26 tiles 661 -> ~, 25 tiles 705 us)
    constexpr int NKE = (NKT + 1) & ~1;
    const size_t lds_a = dq_lds<HD, NKT>(bl), lds_b = dkv_lds<HD, NKE>(bl);
    if (lds_a > MAX_LDS || lds_b > MAX_LDS) return CLV_ERR_UNSUPPORTED;
    if (G.g.mode == 1 && G.drop_thresh) return CLV_ERR_UNSUPPORTED;
    set_attrs<HD, NKT>();
    const int nblk = G.g.groups * G.g.nH * G.tsplit * G.nparts;
    int rc = CLV_OK;
    bf16_t* dq_out = (bf16_t*)dq;
    bf16_t* dk_out = (bf16_t*)dk;
    bf16_t* dv_out = (bf16_t*)dv;
    if (G.nparts > 1) {                                      // partial dq / dk / dv: scratch [part][dq | dk | dv][tokens][C]
        const int looped = (G.g.N + 15) / 16;
        if (!G.g.work || (NKT + DKV_THREADS(NKT) / 64 - 1) / (DKV_THREADS(NKT) / 64) * (DKV_THREADS(NKT) / 64) * G.tsplit < looped ||
            (NKE + DKV_THREADS(NKE) / 64 - 1) / (DKV_THREADS(NKE) / 64) * (DKV_THREADS(NKE) / 64) * G.tsplit < looped)
            return CLV_ERR_UNSUPPORTED;
        const int64_t tc = (int64_t)G.g.groups * G.g.N * G.g.nH * G.g.hd;
        dq_out = reinterpret_cast<bf16_t*>(G.g.work);
        dk_out = dq_out + tc;
        dv_out = dq_out + 2 * tc;
    }
    // stage masks of the one-kernel form: 5 = the kernel alone (dQ + dK / dV + dS scratch), 2 = the table gradient, 7 = both
    const bool one = one_eligible<HD, NKT>(G, bias != nullptr) && (stages & 5) == 5;
    auto launch_one = [&](const Geom& Gx, int groups) {
        if constexpr (HD == 32 && (NKT == 13 || NKT == 25))
            attn_bwd_one_kernel<HD, NKT><<<dim3(groups * G.g.nH), dim3((ONE_CWAVES(NKT) + ONE_SWAVES(NKT)) * 64), one_lds<HD, NKT>(bl), st>>>(
                (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)dout, lse, bias, rid,
                dq_out, dk_out, dv_out, (bf16_t*)(bias ? work : nullptr), dsum, Gx);
    };
    const int64_t E = (int64_t)G.g.nH * ((G.g.N + 15) / 16) * NKT * 64;
    // (the partial sums live behind the scratch, or — so that a caller who gathers later can release the scratch — in the
    // geometry's own `work` buffer: clv_attn_dbias_partial_bytes)
    float* partial = !bias ? nullptr
                     : G.g.work ? reinterpret_cast<float*>(G.g.work)
                                : reinterpret_cast<float*>(reinterpret_cast<char*>(work) + ds_scratch_bytes(G, NKT));
    // >= 16 groups per slice: the fp32 partial tables cost 2 x 16 B per 4 scores and slice, i.e. as much as the
    // bf16 scratch itself once a slice covers only 4 groups
    const int splits = dbias_splits(G);
    // The dS scratch of a stage-0 block is 266 MB at 16 clips: written by the dQ kernel, read once by the streaming sum —
    // a 0.5 GB HBM round trip per block that exists only to feed the table gradient.  Run the pair in CHUNKS of groups that
    // reuse one scratch region small enough for the Infinity Cache (256 MB): the sum then reads what the dQ kernel just
    // wrote from the cache, and the next chunk overwrites the same lines before they are ever written back.
    static const int chunk_mb = getenv("CLV_DBIAS_CHUNK_MB") ? atoi(getenv("CLV_DBIAS_CHUNK_MB")) : 128;      // same-box A/B: 12.52 -> 12.46 ms per step
    int nch = 1;
    if (bias && (stages & 7) == 7 && chunk_mb > 0 && G.nparts == 1) {
        const int64_t per_group = E * 2;                       // scratch bytes per group
        while (nch < 8 && (G.g.groups / nch) * per_group > (int64_t)chunk_mb << 20 && G.g.groups % (2 * nch) == 0 &&
               splits % (2 * nch) == 0)
            nch *= 2;
    }
    if (nch > 1) {
        const int cg = G.g.groups / nch, spc = splits / nch;
        for (int cix = 0; cix < nch; ++cix) {
            Geom Cg = G;
            Cg.grp0 = cix * cg;
            const int nb = cg * G.g.nH * G.tsplit;
            if (one) launch_one(Cg, cg);
            else
            CLV_PICK(attn_bwd_dq_kernel, <<<dim3(nb), dim3(DKV_THREADS(NKT)), lds_a, st>>>((const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)dout, lse, bias, rid, kmask, dq_out,
                     (bf16_t*)work, dsum, seed, Cg));
            hipLaunchKernelGGL(dbias_sum_kernel, dim3((unsigned)((E / 2 + 255) / 256), spc), dim3(256), 0, st,
                               (const bf16_t*)work, reinterpret_cast<float4*>(partial + (int64_t)cix * spc * E * 4), cg, E / 2);
        }
        rc = clv_check_launch();
        if (rc) return rc;
    } else if (one) {
        launch_one(G, G.g.groups);
        rc = clv_check_launch();
        if (rc) return rc;
    } else if (stages & 1) {
        CLV_PICK(attn_bwd_dq_kernel, <<<dim3(nblk), dim3(DKV_THREADS(NKT)), lds_a, st>>>((const bf16_t*)q, (const bf16_t*)k,
                 (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)dout, lse, bias, rid, kmask, dq_out,
                 (bf16_t*)(bias ? work : nullptr), dsum, seed, G));
        rc = clv_check_launch();
        if (rc) return rc;
    }
    if (bias && (stages & 2)) {
        float* dense = partial + (int64_t)splits * E * 4;
        if (nch == 1)
            hipLaunchKernelGGL(dbias_sum_kernel, dim3((unsigned)((E / 2 + 255) / 256), splits), dim3(256), 0, st,
                               (const bf16_t*)work, reinterpret_cast<float4*>(partial), G.g.groups, E / 2);
        (void)dense;
        if (stages & 8) {
            // the caller gathers later (clv_attn_dbias_gather_batch): nothing more to launch here
        } else if (G.g.dbias_index) {
            hipLaunchKernelGGL(dbias_gather_tab_kernel, dim3((G.tbn * G.g.nH + 3) / 4), dim3(256), 0, st, partial, dbias,
                               G.g.dbias_index, NKT, G, splits, E * 4, G.tb0, G.tbn);
        }
        else
            hipLaunchKernelGGL(dbias_gather_kernel, dim3((G.tlen * G.g.nH + 3) / 4), dim3(256), 0, st, partial, dbias, NKT, G,
                               splits, E * 4);
        rc = clv_check_launch();
        if (rc) return rc;
    }
    if ((stages & 4) && !one) {
        CLV_PICK_N(attn_bwd_dkv_kernel, NKE, <<<dim3(nblk), dim3(DKV_THREADS(NKE)), lds_b, st>>>((const bf16_t*)q, (const bf16_t*)k,
                 (const bf16_t*)v, (const bf16_t*)dout, lse, dsum, bias, rid, kmask, dk_out, dv_out, seed, G));
        rc = clv_check_launch();
        if (rc) return rc;
        if (G.nparts > 1) {
            const int64_t items = 3 * (int64_t)G.g.groups * G.g.N * (G.g.nH * G.g.hd / 8);
            hipLaunchKernelGGL(seq_combine_bwd_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<const bf16_t*>(G.g.work), (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, G);
            rc = clv_check_launch();
        }
    }
    return rc;
}

// smallest instantiated key-tile count >= need
int pick_nkt(int N) {
    const int need = (N + 15) / 16;
    // 13 / 25: the 196- and 392-token windows (4 x 7 x 7, 8 x 7 x 7), 15: the 228-token fusion sequence — no padding-only tiles
    const int opts[] = {2, 8, 13, 14, 15, 16, 25, 28};
    for (int o : opts) if (o >= need) return o;
    return -1;
}

#define DISPATCH_NKT(HDV, FN, ...)                                   \
    switch (nkt) {                                                   \
        case 2: return FN<HDV, 2>(__VA_ARGS__);                      \
        case 8: return FN<HDV, 8>(__VA_ARGS__);                      \
        case 13: return FN<HDV, 13>(__VA_ARGS__);                    \
        case 14: return FN<HDV, 14>(__VA_ARGS__);                    \
        case 15: return FN<HDV, 15>(__VA_ARGS__);                    \
        case 16: return FN<HDV, 16>(__VA_ARGS__);                    \
        case 25: return FN<HDV, 25>(__VA_ARGS__);                    \
        case 28: return FN<HDV, 28>(__VA_ARGS__);                    \
        default: return CLV_ERR_UNSUPPORTED;                         \
    }

}  // namespace

extern "C" int clv_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                            const float* bias, const int32_t* rid, const float* kmask, const void* seed,
                            const ClvAttnGeom* geom, void* stream) {
    Geom G;
    if (!q || !k || !v || !o || !lse || !make_geom(geom, G)) return CLV_ERR_ARG;
    if (G.drop_thresh && !seed) return CLV_ERR_ARG;
    const unsigned long long* sp = (const unsigned long long*)seed;
    if (bias && (G.g.mode != 1 || G.tlen == 0)) return CLV_ERR_ARG;
    if (rid && G.g.mode != 1) return CLV_ERR_ARG;
    const int nkt = pick_nkt(G.nparts > 1 ? G.pt16 : G.g.N);
    hipStream_t st = (hipStream_t)stream;
    if (G.g.hd == 16) { DISPATCH_NKT(16, launch_fwd, q, k, v, o, lse, bias, rid, kmask, sp, G, st) }
    if (G.g.hd == 32) { DISPATCH_NKT(32, launch_fwd, q, k, v, o, lse, bias, rid, kmask, sp, G, st) }
    DISPATCH_NKT(64, launch_fwd, q, k, v, o, lse, bias, rid, kmask, sp, G, st)
}

extern "C" int64_t clv_attn_bwd_work_bytes(const ClvAttnGeom* geom) {
    Geom G;
    if (!make_geom(geom, G) || G.g.mode != 1 || G.tlen == 0) return 0;
    const int nkt = pick_nkt(G.g.N);
    if (nkt < 0) return 0;
    return ds_scratch_bytes(G, nkt) + (int64_t)(DBIAS_SPLITS + 1) * G.g.nH * ((G.g.N + 15) / 16) * nkt * 256 * 4;   // + fp32 partials, dense
}

extern "C" int64_t clv_attn_dbias_partial_bytes(const ClvAttnGeom* geom) {
    Geom G;
    if (!make_geom(geom, G) || G.g.mode != 1 || G.tlen == 0) return 0;
    const int nkt = pick_nkt(G.g.N);
    if (nkt < 0) return 0;
    return (int64_t)dbias_splits(G) * G.g.nH * ((G.g.N + 15) / 16) * nkt * 256 * 4;
}

extern "C" int clv_attn_bwd_one_kernel(const ClvAttnGeom* geom) {
    Geom G;
    if (!make_geom(geom, G) || G.g.mode != 1 || G.tlen == 0 || G.g.hd != 32) return 0;
    const int nkt = pick_nkt(G.g.N);
    DISPATCH_NKT(32, query_one, G)
}

extern "C" int64_t clv_attn_seq_work_bytes(const ClvAttnGeom* geom) {
    Geom G;
    if (!make_geom(geom, G) || G.nparts == 1) return 0;
    // forward: partial o + lse per part; backward: partial dq / dk / dv per part (the larger of the two)
    const int64_t tc = (int64_t)G.g.groups * G.g.N * G.g.nH * G.g.hd;
    const int64_t fwd = G.nparts * (tc * 2 + G.lse_ps * 4), bwd = G.nparts * 3 * tc * 2;
    return fwd > bwd ? fwd : bwd;
}

extern "C" int clv_attn_seq_max_keys(void) { return 2 * SEQ_ONE_PART_TILES * 16; }

extern "C" int64_t clv_attn_dbias_index_count(const ClvAttnGeom* geom) {
    Geom G;
    if (!make_geom(geom, G) || G.g.mode != 1 || G.tlen == 0) return 0;
    return (int64_t)G.tlen * pick_nkt(G.g.N) * 16;
}

extern "C" int clv_attn_dbias_index(const ClvAttnGeom* geom, int32_t* out, void* stream) {
    Geom G;
    if (!out || !make_geom(geom, G) || G.g.mode != 1 || G.tlen == 0) return CLV_ERR_ARG;
    const int nkt = pick_nkt(G.g.N);
    const int64_t n = (int64_t)G.tlen * nkt * 16;
    hipLaunchKernelGGL(dbias_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, nkt, G);
    return clv_check_launch();
}

extern "C" int clv_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                            const float* lse, const float* bias, const int32_t* rid, const float* kmask,
                            void* dq, void* dk, void* dv, float* dbias, float* dsum, void* work,
                            const void* seed, int32_t stages, const ClvAttnGeom* geom, void* stream) {
    Geom G;
    const unsigned long long* sp = (const unsigned long long*)seed;
    if (stages == 0) stages = 7;
    if (!q || !k || !v || !o || !dout || !lse || !dq || !dk || !dv || !dsum || !make_geom(geom, G)) return CLV_ERR_ARG;
    if (bias && (G.g.mode != 1 || G.tlen == 0 || !dbias || !work)) return CLV_ERR_ARG;
    if (rid && G.g.mode != 1) return CLV_ERR_ARG;
    if (G.drop_thresh && !seed) return CLV_ERR_ARG;
    const int nkt = pick_nkt(G.nparts > 1 ? G.pt16 : G.g.N);
    hipStream_t st = (hipStream_t)stream;
    if (G.g.hd == 16) { DISPATCH_NKT(16, launch_bwd, q, k, v, o, dout, lse, bias, rid, kmask, dq, dk, dv, dbias, dsum, work, sp, stages, G, st) }
    if (G.g.hd == 32) { DISPATCH_NKT(32, launch_bwd, q, k, v, o, dout, lse, bias, rid, kmask, dq, dk, dv, dbias, dsum, work, sp, stages, G, st) }
    DISPATCH_NKT(64, launch_bwd, q, k, v, o, dout, lse, bias, rid, kmask, dq, dk, dv, dbias, dsum, work, sp, stages, G, st)
}


extern "C" int clv_attn_dbias_gather_entry(const ClvAttnGeom* geom, void* work, float* dbias, ClvDbiasGather* out) {
    Geom G;
    if (!geom || !work || !dbias || !out || !make_geom(geom, G) || G.g.mode != 1 || G.tlen == 0 || !G.g.dbias_index)
        return CLV_ERR_ARG;
    const int nkt = pick_nkt(G.g.N);
    if (nkt < 0) return CLV_ERR_UNSUPPORTED;
    const int64_t E = (int64_t)G.g.nH * ((G.g.N + 15) / 16) * nkt * 64;
    out->partial = G.g.work ? reinterpret_cast<char*>(G.g.work) : reinterpret_cast<char*>(work) + ds_scratch_bytes(G, nkt);
    out->dtable = dbias;
    out->index = G.g.dbias_index;
    out->split_stride = E * 4;
    out->nkt = nkt;
    out->nH = G.g.nH;
    out->N = G.g.N;
    out->nsplit = dbias_splits(G);
    out->slot0 = G.tb0;
    out->nslots = G.tbn;
    out->block_begin = 0;
    out->pad = 0;
    return CLV_OK;
}

extern "C" int clv_attn_dbias_gather_batch(const ClvDbiasGather* entries, int32_t n, void* stream) {
    if (!entries || n <= 0 || n > CLV_DBIAS_GATHER_MAX) return CLV_ERR_ARG;
    static_assert(sizeof(ClvDbiasGather) == 64, "ClvDbiasGather layout is part of the ABI");
    GatherTable tab;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        ClvDbiasGather en = entries[i];
        if (!en.partial || !en.dtable || !en.index || en.nsplit <= 0 || en.nslots <= 0 || en.nH <= 0) return CLV_ERR_ARG;
        tab.e[i] = en;
    }
    tab.n = n;
    static const bool presum = !getenv("CLV_DBIAS_PRESUM") || atoi(getenv("CLV_DBIAS_PRESUM")) != 0;
    static const int hg_max = getenv("CLV_DBIAS_HEADS_PER_WAVE") ? atoi(getenv("CLV_DBIAS_HEADS_PER_WAVE")) : 4;
    for (int i = 0; i < n; ++i) {                              // heads per wave (field `pad`): only on summed slices
        ClvDbiasGather& en = tab.e[i];
        int hg = 1;
        if (presum || en.nsplit == 1)
            for (int c = 4; c >= 2; --c)
                if (c <= hg_max && en.nH % c == 0) { hg = c; break; }
        en.pad = hg;
        en.block_begin = blocks;
        blocks += (en.nslots * (en.nH / hg) + 3) / 4;
    }
    if (presum) {
        GatherTable st = tab;
        int sblocks = 0;
        bool any = false;
        for (int i = 0; i < n; ++i) {
            if ((st.e[i].split_stride & 3) || (reinterpret_cast<uintptr_t>(st.e[i].partial) & 15)) return CLV_ERR_ARG;
            st.e[i].block_begin = sblocks;
            sblocks += (int)(((st.e[i].split_stride >> 2) + 255) / 256);
            any |= st.e[i].nsplit > 1;
            tab.e[i].nsplit = 1;                                // the gather below reads slice 0 = the sum
        }
        if (any) {
            hipLaunchKernelGGL(dbias_split_sum_batch_kernel, dim3((unsigned)sblocks), dim3(256), 0, (hipStream_t)stream, st);
            const int rc = clv_check_launch();
            if (rc) return rc;
        }
    }
    hipLaunchKernelGGL(dbias_gather_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab);
    return clv_check_launch();
}

extern "C" int clv_attn_f32_fwd(const float* q, const float* k, const float* v, float* o, const float* bias,
                                const int32_t* rid, const float* kmask, const ClvAttnGeom* geom, int32_t round_p,
                                void* stream) {
    Geom G;
    if (!q || !k || !v || !o || !make_geom(geom, G)) return CLV_ERR_ARG;
    if (bias && (G.g.mode != 1 || G.tlen == 0)) return CLV_ERR_ARG;
    if (rid && G.g.mode != 1) return CLV_ERR_ARG;
    if (G.g.N > F32_MAXT * 64 || G.drop_thresh) return CLV_ERR_UNSUPPORTED;      // forward-only, eval-mode kernel
    const size_t lds = (size_t)G.g.N * (4 * 4 + 4 * 4);
    const int pairs = G.g.groups * G.g.nH;
    int ysplit = 1;
    while (pairs * ysplit < 2048 && ysplit * 8 <= G.g.N) ysplit *= 2;
    hipLaunchKernelGGL(attn_f32_fwd_kernel, dim3(pairs, ysplit), dim3(256), lds, (hipStream_t)stream, q, k, v, o, bias, rid,
                       kmask, G, round_p);
    return clv_check_launch();
}

extern "C" int64_t clv_attn_f32_bwd_work_floats(const ClvAttnGeom* geom) {
    Geom G;
    if (!make_geom(geom, G)) return 0;
    return (int64_t)2 * G.g.groups * G.g.nH * G.g.N;
}

extern "C" int clv_attn_f32_bwd(const float* q, const float* k, const float* v, const float* dout, const float* bias,
                                const int32_t* rid, const float* kmask, float* dq, float* dk, float* dv, float* dbias,
                                float* work, const ClvAttnGeom* geom, void* stream) {
    Geom G;
    if (!q || !k || !v || !dout || !dq || !dk || !dv || !work || !make_geom(geom, G)) return CLV_ERR_ARG;
    if (bias && (G.g.mode != 1 || G.tlen == 0)) return CLV_ERR_ARG;
    if (rid && G.g.mode != 1) return CLV_ERR_ARG;
    if (G.g.N > F32_MAXT * 64 || G.drop_thresh) return CLV_ERR_UNSUPPORTED;      // eval-mode arithmetic, as the forward
    const size_t lds = (size_t)G.g.N * (4 * 4 + 8 * 4);
    const int pairs = G.g.groups * G.g.nH;
    int ysplit = 1;
    while (pairs * ysplit < 2048 && ysplit * 8 <= G.g.N) ysplit *= 2;
    float* lse_w = work;
    float* delta_w = work + (int64_t)pairs * G.g.N;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((attn_f32_bwd_kernel<false>), dim3(pairs, ysplit), dim3(256), lds, st, q, k, v, dout, bias, rid, kmask,
                       dq, dk, dv, dbias, lse_w, delta_w, G);
    int rc = clv_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL((attn_f32_bwd_kernel<true>), dim3(pairs, ysplit), dim3(256), lds, st, q, k, v, dout, bias, rid, kmask,
                       dq, dk, dv, dbias, lse_w, delta_w, G);
    return clv_check_launch();
}
