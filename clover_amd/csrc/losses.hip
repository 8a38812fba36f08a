// Fused losses of the Clover pre-training step (fp32 math, as the reference's @force_fp32):
//   * focal masked-LM cross entropy over the vocabulary (one pass over the logits);
//   * exclusive tri-modal InfoNCE + margin ranking on the all-gathered embeddings
//     (normalise -> f32-MFMA similarity GEMMs -> one-block loss -> analytic backward).
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

// =========================================================================== focal CE
constexpr int FC_THREADS = 256;

template <bool BF16>
__device__ __forceinline__ float ld_logit(const void* p, int64_t i) {
    if (BF16) return bf2f(reinterpret_cast<const bf16_t*>(p)[i]);
    return reinterpret_cast<const float*>(p)[i];
}

__device__ __forceinline__ float block_reduce(float v, float* sh, bool is_max) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    float r = sh[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
    return r;
}

template <bool BF16>
__global__ void __launch_bounds__(FC_THREADS) focal_fwd_kernel(
    const void* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ row_ce,
    float* __restrict__ row_lse, float* __restrict__ sum_acc, float* __restrict__ cnt_acc, int V, int64_t ld, float gamma) {
    __shared__ float sh[8];
    const int64_t row = blockIdx.x;
    const int64_t lab = labels[row];
    if (lab < 0) {                       // label == -100: not a masked position
        if (threadIdx.x == 0) { row_ce[row] = 0.f; row_lse[row] = 0.f; }
        return;
    }
    const int64_t base = row * ld;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < V; j += FC_THREADS) m = fmaxf(m, ld_logit<BF16>(logits, base + j));
    m = block_reduce(m, sh, true);
    float s = 0.f;
    for (int j = threadIdx.x; j < V; j += FC_THREADS) s += __expf(ld_logit<BF16>(logits, base + j) - m);
    s = block_reduce(s, sh, false);
    if (threadIdx.x == 0) {
        const float lse = m + __logf(s);
        const float ce = lse - ld_logit<BF16>(logits, base + lab);
        const float pt = __expf(-ce);
        row_ce[row] = ce;
        row_lse[row] = lse;
        atomicAdd(sum_acc, powf(1.f - pt, gamma) * ce);
        atomicAdd(cnt_acc, 1.0f);
    }
}

__global__ void focal_finish_kernel(float* __restrict__ loss, const float* __restrict__ count) {
    loss[0] = loss[0] / count[0];        // mean over the masked rows (NaN when there are none, as torch)
}

template <bool BF16>
__global__ void __launch_bounds__(FC_THREADS) focal_bwd_kernel(
    const void* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ row_ce,
    const float* __restrict__ row_lse, const float* __restrict__ count, const float* __restrict__ dloss,
    void* __restrict__ dlogits, int V, int64_t ld, float gamma) {
    const int64_t row = blockIdx.x;
    const int64_t lab = labels[row];
    const int64_t base = row * ld;
    float coef = 0.f, lse = 0.f;
    if (lab >= 0) {
        const float ce = row_ce[row], pt = __expf(-ce), om = 1.f - pt;
        // d/dce [(1-pt)^g * ce] = g (1-pt)^(g-1) pt ce + (1-pt)^g
        const float dl = (gamma == 0.f) ? 1.f : gamma * powf(om, gamma - 1.f) * pt * ce + powf(om, gamma);
        coef = dloss[0] * dl / count[0];
        lse = row_lse[row];
    }
    for (int j = threadIdx.x; j < (int)ld; j += FC_THREADS) {     // columns [V, ld) are padding: zero
        float g = 0.f;
        if (lab >= 0 && j < V) {
            const float p = __expf(ld_logit<BF16>(logits, base + j) - lse);
            g = coef * (p - (j == lab ? 1.f : 0.f));
        }
        if (BF16) reinterpret_cast<bf16_t*>(dlogits)[base + j] = f2bf(g);
        else reinterpret_cast<float*>(dlogits)[base + j] = g;
    }
}

// bf16 logits with an even vocabulary (30522): a row is 4-byte aligned, so a thread moves PAIRS (one 4-byte access), 1024
// threads per row, and the forward keeps a running (max, sum) per thread — one pass over the row, 15 independent loads
// per thread instead of two passes of 120 dependent 2-byte ones (43 -> ~10 us for the ~40 masked rows of a step, which
// sit at the tail of the forward graph and at the head of the backward graph).
constexpr int FC_WIDE = 1024;

__global__ void __launch_bounds__(FC_WIDE) focal_fwd_pairs_kernel(
    const unsigned* __restrict__ logits2, const int64_t* __restrict__ labels, float* __restrict__ row_ce,
    float* __restrict__ row_lse, float* __restrict__ sum_acc, float* __restrict__ cnt_acc, int V, int64_t ld, float gamma) {
    __shared__ float shm[FC_WIDE / 64], shs[FC_WIDE / 64];
    const int64_t row = blockIdx.x;
    const int64_t lab = labels[row];
    if (lab < 0) {
        if (threadIdx.x == 0) { row_ce[row] = 0.f; row_lse[row] = 0.f; }
        return;
    }
    const int V2 = V >> 1;
    const unsigned* r2 = logits2 + row * (ld >> 1);
    float m = -INFINITY, sum = 0.f;
    for (int j = threadIdx.x; j < V2; j += FC_WIDE) {
        const unsigned w = r2[j];
        const float a = bf2f((bf16_t)(w & 0xffff)), b = bf2f((bf16_t)(w >> 16));
        const float mn = fmaxf(m, fmaxf(a, b));
        sum = sum * __expf(m - mn) + __expf(a - mn) + __expf(b - mn);     // first trip: 0 * exp(-inf) = 0
        m = mn;
    }
    // (m, sum) pairs: wave, then block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float wm = wave_max(m);
    sum = wave_sum(m == -INFINITY ? 0.f : sum * __expf(m - wm));          // a lane (or a whole wave) without elements
    if (lane == 0) { shm[wv] = wm; shs[wv] = sum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bm = shm[0];
        for (int w = 1; w < FC_WIDE / 64; ++w) bm = fmaxf(bm, shm[w]);
        float bs = 0.f;
        for (int w = 0; w < FC_WIDE / 64; ++w)
            if (shm[w] != -INFINITY) bs += shs[w] * __expf(shm[w] - bm);
        const float lse = bm + __logf(bs);
        const unsigned w = r2[lab >> 1];
        const float ce = lse - bf2f((bf16_t)((lab & 1) ? (w >> 16) : (w & 0xffff)));
        const float pt = __expf(-ce);
        row_ce[row] = ce;
        row_lse[row] = lse;
        atomicAdd(sum_acc, powf(1.f - pt, gamma) * ce);
        atomicAdd(cnt_acc, 1.0f);
    }
}

__global__ void __launch_bounds__(FC_WIDE) focal_bwd_pairs_kernel(
    const unsigned* __restrict__ logits2, const int64_t* __restrict__ labels, const float* __restrict__ row_ce,
    const float* __restrict__ row_lse, const float* __restrict__ count, const float* __restrict__ dloss,
    unsigned* __restrict__ dlogits2, int V, int64_t ld, float gamma) {
    const int64_t row = blockIdx.x;
    const int64_t lab = labels[row];
    const int V2 = V >> 1, L2 = (int)(ld >> 1);
    unsigned* d2 = dlogits2 + row * L2;
    if (lab < 0) {
        for (int j = threadIdx.x; j < L2; j += FC_WIDE) d2[j] = 0u;
        return;
    }
    for (int j = V2 + threadIdx.x; j < L2; j += FC_WIDE) d2[j] = 0u;      // padding columns [V, ld)
    const float ce = row_ce[row], pt = __expf(-ce), om = 1.f - pt;
    const float dl = (gamma == 0.f) ? 1.f : gamma * powf(om, gamma - 1.f) * pt * ce + powf(om, gamma);
    const float coef = dloss[0] * dl / count[0], lse = row_lse[row];
    const unsigned* r2 = logits2 + row * L2;
    const int labp = (int)(lab >> 1);
    for (int j = threadIdx.x; j < V2; j += FC_WIDE) {
        const unsigned w = r2[j];
        float ga = coef * __expf(bf2f((bf16_t)(w & 0xffff)) - lse), gb = coef * __expf(bf2f((bf16_t)(w >> 16)) - lse);
        if (j == labp) {
            if (lab & 1) gb -= coef;
            else ga -= coef;
        }
        d2[j] = pack2bf(ga, gb);
    }
}

// =========================================================================== InfoNCE
// work layout (floats):
//   en   [4][G][Dm]   normalised embeddings
//   enT  [Dm][4][G]   their transpose (contraction index (k, j) contiguous)
//   invn [4][G]       1/max(|e|, eps)
//   sim  [3][G][G]    en0 . en_{k+1}^T / temperature
//   lser [3][G]       row log-sum-exp of the three exclusive [G,3G] rows
//   lsec [3][G]       column log-sum-exp (t2v)
//   dsim [G][3][G]    d loss / d sim, row-major in (i, k, j)
//   dsimT[3][G][G]    transposed per block: [k][j][i]
//   den  [4][G][Dm]   d loss / d normalised embeddings
//   acc  [2]
struct NceWork {
    float *en, *enT, *invn, *sim, *lser, *lsec, *dsim, *dsimT, *den, *acc;
};
__host__ __device__ inline int64_t nce_work_floats(int G, int Dm) {
    return (int64_t)4 * G * Dm * 3 + (int64_t)4 * G + (int64_t)3 * G * G * 3 + (int64_t)6 * G + 16;
}
__host__ __device__ inline NceWork nce_carve(float* w, int G, int Dm) {
    NceWork W;
    W.en = w; w += (int64_t)4 * G * Dm;
    W.enT = w; w += (int64_t)4 * G * Dm;
    W.den = w; w += (int64_t)4 * G * Dm;
    W.invn = w; w += 4 * G;
    W.sim = w; w += (int64_t)3 * G * G;
    W.dsim = w; w += (int64_t)3 * G * G;
    W.dsimT = w; w += (int64_t)3 * G * G;
    W.lser = w; w += 3 * G;
    W.lsec = w; w += 3 * G;
    W.acc = w;
    return W;
}

// one wave per row: en = e / max(|e|, 1e-8)   (cos_norm, contrastive_loss.py:20-25)
__device__ __forceinline__ void nce_normalize_body(const float* e0, const float* e1, const float* e2, const float* e3,
                                                   const NceWork& W, int G, int Dm, int ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 4 * G) return;
    const int k = row / G, i = row - k * G;
    const float* e = (k == 0 ? e0 : k == 1 ? e1 : k == 2 ? e2 : e3) + (int64_t)i * ld;
    float s = 0.f;
    for (int d = lane; d < Dm; d += 64) s += e[d] * e[d];
    const float nrm = sqrtf(wave_sum(s));
    const float inv = 1.0f / fmaxf(nrm, 1e-8f);
    if (lane == 0) W.invn[row] = inv;
    for (int d = lane; d < Dm; d += 64) {
        const float v = e[d] * inv;
        W.en[(int64_t)row * Dm + d] = v;
        W.enT[((int64_t)d * 4 + k) * G + i] = v;
    }
}
__global__ void __launch_bounds__(256) nce_normalize_kernel(const float* e0, const float* e1, const float* e2,
                                                            const float* e3, NceWork W, int G, int Dm, int ld) {
    nce_normalize_body(e0, e1, e2, e3, W, G, Dm, ld);
}

// The recognizer evaluates the loss TWICE per step on slots of one packed [G, k, Dm] tensor (video -> text and text -> video,
// multimodal_transformer_pretrain.py:151,161): the *_pair kernels run both evaluations ("directions") in the same launches —
// blockIdx.y (or the batch index of the GEMMs) is the direction, each with its own work area — so that the loss section
// is 3 + 4 dependent launches instead of 6 + 8 (+ the zero fills and the add of the two gradients).
struct NcePair {
    NceWork w[2];
    const float* e[2][4];
    const float* dout[4];      // backward: device scalars, the upstream gradients of {nce_0, rank_0, nce_1, rank_1} (null: 0)
    int slot[2][4];
};
__global__ void __launch_bounds__(256) nce_pair_normalize_kernel(NcePair P, int G, int Dm, int ld) {
    const int d = blockIdx.y;
    nce_normalize_body(P.e[d][0], P.e[d][1], P.e[d][2], P.e[d][3], P.w[d], G, Dm, ld);
}

// C[i][j] (ldc) = alpha * sum_k A[i][k] * B[j][k]   — exact-f32 MFMA 16x16x4, one 16x16 tile per workgroup.
// K multiple of 16 is NOT required (tail guarded); rows/cols guarded.  NW waves split the contraction (partial tiles meet in
// LDS): the similarity GEMMs have G x G outputs — a handful of tiles — and Dm = 768 to contract: one wave per tile walked 48
// dependent load rounds (20 us); the gradient GEMMs contract over <= 3 G and keep one wave.
template <int NW>
__global__ void __launch_bounds__(64 * NW) sgemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int M, int N, int K, int lda, int ldb,
                                                      int ldc, int64_t sA, int64_t sB, int64_t sC, float alpha,
                                                      int zb, int64_t s2) {
    // batch index z = z1 * zb + z0: operand offsets (sA, sB, sC) * z0 + s2 * z1 (the pair form: z1 = direction, one work
    // area of s2 floats each)
    __shared__ float red[NW > 1 ? NW - 1 : 1][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane >> 4, lr = lane & 15;
    const int z1 = blockIdx.z / zb, z0 = blockIdx.z - z1 * zb;
    A += sA * z0 + s2 * z1;
    B += sB * z0 + s2 * z1;
    C += sC * z0 + s2 * z1;
    const int i = blockIdx.y * 16 + lr, j = blockIdx.x * 16 + lr;
    const float* ap = A + (int64_t)(i < M ? i : 0) * lda;
    const float* bp = B + (int64_t)(j < N ? j : 0) * ldb;
    const bool av = i < M, bv = j < N;
    const int kchunk = ((K + NW * 16 - 1) / (NW * 16)) * 16;       // per wave, a multiple of 16
    const int kb = wave * kchunk, ke = min(K, kb + kchunk);
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = kb; k0 < ke; k0 += 16) {
        float a[4], b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + lg * 4 + e;
            a[e] = (av && k < ke) ? ap[k] : 0.f;
            b[e] = (bv && k < ke) ? bp[k] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
    }
    if (NW > 1) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][lane][r] = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < NW - 1; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += red[w][lane][r];
    }
    // acc[r] = C[row blockIdx.y*16 + lg*4 + r][col blockIdx.x*16 + lr]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ci = blockIdx.y * 16 + lg * 4 + r;
        if (ci < M && j < N) C[(int64_t)ci * ldc + j] = alpha * acc[r];
    }
}

__device__ __forceinline__ float nce_entry(const float* sim, int G, int kx, int i, int k, int j) {
    // value of column (k, j) in the exclusive row of variant kx for sample i
    // (contrastive_loss.py:130-132: other blocks' diagonals become x - (x + 10000))
    const float x = sim[((int64_t)k * G + i) * G + j];
    if (k != kx && j == i) return x - (x + 10000.f);
    return x;
}

// single block: all log-sum-exps, the two losses.
__device__ __forceinline__ void nce_loss_body(const NceWork& W, float* __restrict__ out, int G, float margin) {
    __shared__ float sh[16];
    const float* sim = W.sim;
    float part_v = 0.f, part_t = 0.f, part_r = 0.f;
    // 3G exclusive rows + 3G columns, one thread each (G <= 512 in practice: 3G*G work per thread is tiny)
    for (int t = threadIdx.x; t < 6 * G; t += blockDim.x) {
        if (t < 3 * G) {
            const int kx = t / G, i = t - kx * G;
            float m = -INFINITY;
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < G; ++j) m = fmaxf(m, nce_entry(sim, G, kx, i, k, j));
            float s = 0.f;
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < G; ++j) s += __expf(nce_entry(sim, G, kx, i, k, j) - m);
            const float lse = m + __logf(s);
            W.lser[t] = lse;
            part_v += sim[((int64_t)kx * G + i) * G + i] - lse;          // diag of the own block
        } else {
            const int u = t - 3 * G, k = u / G, j = u - k * G;
            float m = -INFINITY;
            for (int i = 0; i < G; ++i) m = fmaxf(m, sim[((int64_t)k * G + i) * G + j]);
            float s = 0.f;
            for (int i = 0; i < G; ++i) s += __expf(sim[((int64_t)k * G + i) * G + j] - m);
            const float lse = m + __logf(s);
            W.lsec[u] = lse;
            part_t += sim[((int64_t)k * G + j) * G + j] - lse;
        }
    }
    for (int i = threadIdx.x; i < G; i += blockDim.x) {
        const float a = sim[((int64_t)0 * G + i) * G + i], b = sim[((int64_t)1 * G + i) * G + i];
        part_r += fmaxf(0.f, -(a - b) + margin);
    }
    const float sv = block_reduce(part_v, sh, false);
    const float stt = block_reduce(part_t, sh, false);
    const float sr = block_reduce(part_r, sh, false);
    if (threadIdx.x == 0) {
        const float loss_v = -(sv / (float)G);
        const float loss_t = -(stt / (float)(3 * G));
        out[0] = loss_v + loss_t;
        out[1] = sr / (float)G;
    }
}
__global__ void __launch_bounds__(1024) nce_loss_kernel(NceWork W, float* __restrict__ out, int G, float margin) {
    nce_loss_body(W, out, G, margin);
}
__global__ void __launch_bounds__(1024) nce_pair_loss_kernel(NcePair P, float* __restrict__ out, int G, float margin) {
    nce_loss_body(P.w[blockIdx.x], out + 2 * blockIdx.x, G, margin);
}

// elementwise: d loss / d sim[k][i][j]
__device__ __forceinline__ void nce_dsim_body(const NceWork& W, const float gn, const float gr, int G, float margin) {
    const int64_t total = (int64_t)3 * G * G;
    const float invG = 1.0f / (float)G, inv3G = 1.0f / (float)(3 * G);
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(idx / ((int64_t)G * G));
        const int rem = (int)(idx - (int64_t)k * G * G);
        const int i = rem / G, j = rem - i * G;
        const float x = W.sim[idx];
        float g = 0.f;
        // v2t: three exclusive rows of sample i
        for (int kx = 0; kx < 3; ++kx) {
            if (k != kx && j == i) continue;                    // excluded entry: x-(x+1e4) has zero slope
            const float p = __expf(x - W.lser[kx * G + i]);
            g += -invG * gn * ((k == kx && j == i ? 1.f : 0.f) - p);
        }
        // t2v: column softmax over i
        g += -inv3G * gn * ((i == j ? 1.f : 0.f) - __expf(x - W.lsec[k * G + j]));
        // ranking: mean_i max(0, margin - (vt_ii - vtm_ii))
        if (i == j && k < 2) {
            const float a = W.sim[((int64_t)0 * G + i) * G + i], b = W.sim[((int64_t)1 * G + i) * G + i];
            if (-(a - b) + margin > 0.f) g += (k == 0 ? -1.f : 1.f) * invG * gr;
        }
        W.dsim[((int64_t)i * 3 + k) * G + j] = g;
        W.dsimT[((int64_t)k * G + j) * G + i] = g;
    }
}
__global__ void __launch_bounds__(256) nce_dsim_kernel(NceWork W, const float* __restrict__ dout, int G, float margin) {
    nce_dsim_body(W, dout[0], dout[1], G, margin);
}
__global__ void __launch_bounds__(256) nce_pair_dsim_kernel(NcePair P, int G, float margin) {
    const float* gn = P.dout[2 * blockIdx.y];
    const float* gr = P.dout[2 * blockIdx.y + 1];
    nce_dsim_body(P.w[blockIdx.y], gn ? *gn : 0.f, gr ? *gr : 0.f, G, margin);
}

// d e = invn * (d en - en * <en, d en>)   (valid while |e| >= eps; below eps: d e = d en * invn)
__global__ void __launch_bounds__(256) nce_norm_bwd_kernel(NceWork W, float* d0, float* d1, float* d2, float* d3,
                                                           int G, int Dm, int ldd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 4 * G) return;
    const int k = row / G, i = row - k * G;
    float* d = (k == 0 ? d0 : k == 1 ? d1 : k == 2 ? d2 : d3) + (int64_t)i * ldd;
    const float* en = W.en + (int64_t)row * Dm;
    const float* de = W.den + (int64_t)row * Dm;
    float s = 0.f;
    for (int c = lane; c < Dm; c += 64) s += en[c] * de[c];
    s = wave_sum(s);
    const float inv = W.invn[row];
    const bool clamped = inv >= 1e8f;
    for (int c = lane; c < Dm; c += 64) d[c] = clamped ? de[c] * inv : inv * (de[c] - en[c] * s);
}

// The pair form writes the WHOLE packed gradient [G][k][Dm]: one wave per (row i, slot s) sums d en over the (direction,
// position) pairs that read slot s (video and text are read by both directions; the map d e = invn (d en - en <en, d en>) is
// linear in d en and en / invn of a slot are the same numbers in both work areas), a slot nobody read gets zeros.
__global__ void __launch_bounds__(256) nce_pair_norm_bwd_kernel(NcePair P, float* __restrict__ dp, int G, int k, int Dm) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= k * G) return;
    const int i = row / k, sl = row - i * k;
    float* d = dp + (int64_t)row * Dm;
    const float* en = nullptr;
    const float* de[2] = {nullptr, nullptr};
    float inv = 0.f;
    int n = 0;
#pragma unroll
    for (int dir = 0; dir < 2; ++dir)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (P.slot[dir][q] == sl && n < 2) {
                const int64_t r = (int64_t)q * G + i;
                if (!n) {
                    en = P.w[dir].en + r * Dm;
                    inv = P.w[dir].invn[r];
                }
                de[n++] = P.w[dir].den + r * Dm;
            }
    if (!n) {
        for (int c = lane; c < Dm; c += 64) d[c] = 0.f;
        return;
    }
    float s = 0.f;
    for (int c = lane; c < Dm; c += 64) s += en[c] * (de[0][c] + (n > 1 ? de[1][c] : 0.f));
    s = wave_sum(s);
    const bool clamped = inv >= 1e8f;
    for (int c = lane; c < Dm; c += 64) {
        const float g = de[0][c] + (n > 1 ? de[1][c] : 0.f);
        d[c] = clamped ? g * inv : inv * (g - en[c] * s);
    }
}


// =========================================================================== NormSoftmaxLoss
// (mmaction/models/losses/contrastive_loss.py:26-68): x = normalise(video) . normalise(text)^T / t,
// loss = -mean_i diag(log_softmax(x, 1)) - mean_j diag(log_softmax(x^T, 1)).
// work layout (floats):
//   en [2][G][Dm], enT [Dm][2][G], den [2][G][Dm], invn [2][G], sim [G][G], dsim [G][G], dsimT [G][G],
//   lser [G], lsec [G]
struct NsWork {
    float *en, *enT, *den, *invn, *sim, *dsim, *dsimT, *lser, *lsec;
};
__host__ __device__ inline int64_t ns_work_floats(int G, int Dm) {
    return (int64_t)2 * G * Dm * 3 + (int64_t)2 * G + (int64_t)G * G * 3 + (int64_t)2 * G + 16;
}
__host__ __device__ inline NsWork ns_carve(float* w, int G, int Dm) {
    NsWork W;
    W.en = w; w += (int64_t)2 * G * Dm;
    W.enT = w; w += (int64_t)2 * G * Dm;
    W.den = w; w += (int64_t)2 * G * Dm;
    W.invn = w; w += 2 * G;
    W.sim = w; w += (int64_t)G * G;
    W.dsim = w; w += (int64_t)G * G;
    W.dsimT = w; w += (int64_t)G * G;
    W.lser = w; w += G;
    W.lsec = w;
    return W;
}

// one wave per row: en = e / max(|e|, eps)  (F.normalize eps 1e-12 :51-52, sim_matrix eps 1e-8 :10-18)
__global__ void __launch_bounds__(256) ns_normalize_kernel(const float* e0, const float* e1, NsWork W, int G, int Dm,
                                                           float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 2 * G) return;
    const int k = row / G, i = row - k * G;
    const float* e = (k == 0 ? e0 : e1) + (int64_t)i * Dm;
    float s = 0.f;
    for (int d = lane; d < Dm; d += 64) s += e[d] * e[d];
    const float nrm = sqrtf(wave_sum(s));
    const float inv = 1.0f / fmaxf(nrm, eps);
    if (lane == 0) W.invn[row] = nrm >= eps ? inv : -inv;          // sign bit marks the clamped rows
    for (int d = lane; d < Dm; d += 64) {
        const float v = e[d] * inv;
        W.en[(int64_t)row * Dm + d] = v;
        W.enT[((int64_t)d * 2 + k) * G + i] = v;
    }
}

// single block: G row + G column log-sum-exps (one thread each), then the loss.
__global__ void __launch_bounds__(1024) ns_loss_kernel(NsWork W, const float* __restrict__ sim, float* __restrict__ out,
                                                       int G) {
    __shared__ float sh[16];
    float part = 0.f;
    for (int t = threadIdx.x; t < 2 * G; t += blockDim.x) {
        const bool col = t >= G;
        const int u = col ? t - G : t;
        const int64_t base = col ? u : (int64_t)u * G, step = col ? G : 1;
        float m = -INFINITY;
        for (int j = 0; j < G; ++j) m = fmaxf(m, sim[base + j * step]);
        float s = 0.f;
        for (int j = 0; j < G; ++j) s += __expf(sim[base + j * step] - m);
        const float lse = m + __logf(s);
        (col ? W.lsec : W.lser)[u] = lse;
        part += sim[(int64_t)u * G + u] - lse;
    }
    const float tot = block_reduce(part, sh, false);
    if (threadIdx.x == 0) out[0] = -(tot / (float)G);
}

__global__ void __launch_bounds__(256) ns_dsim_kernel(NsWork W, const float* __restrict__ sim,
                                                      const float* __restrict__ dout, float* __restrict__ dsim_out,
                                                      int G) {
    const int64_t total = (int64_t)G * G;
    const float g = -dout[0] / (float)G;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx / G), j = (int)(idx - (int64_t)i * G);
        const float x = sim[idx];
        const float d = g * ((i == j ? 2.f : 0.f) - __expf(x - W.lser[i]) - __expf(x - W.lsec[j]));
        dsim_out[idx] = d;
        if (dsim_out == W.dsim) W.dsimT[(int64_t)j * G + i] = d;
    }
}

__global__ void __launch_bounds__(256) ns_norm_bwd_kernel(NsWork W, float* d0, float* d1, int G, int Dm) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= 2 * G) return;
    const int k = row / G, i = row - k * G;
    float* d = (k == 0 ? d0 : d1) + (int64_t)i * Dm;
    const float* en = W.en + (int64_t)row * Dm;
    const float* de = W.den + (int64_t)row * Dm;
    float s = 0.f;
    for (int c = lane; c < Dm; c += 64) s += en[c] * de[c];
    s = wave_sum(s);
    const float inv = W.invn[row];
    const bool clamped = inv < 0.f;
    for (int c = lane; c < Dm; c += 64) d[c] = clamped ? de[c] * -inv : inv * (de[c] - en[c] * s);
}

bool focal_wide() {
    static const bool on = !(getenv("CLV_FOCAL_WIDE") && atoi(getenv("CLV_FOCAL_WIDE")) == 0);   // probe switch
    return on;
}

}  // namespace

static int focal_fwd_impl(const void* logits, int32_t is_bf16, const int64_t* labels, float* row_ce, float* row_lse,
                          float* loss, float* count, int64_t rows, int32_t V, int64_t ld, float gamma, void* stream) {
    if (!logits || !labels || !row_ce || !row_lse || !loss || !count || rows <= 0 || V <= 0 || ld < V) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // (loss, count) double as the {sum, count} accumulators (caller zeroes both); normalised in place
    // by the finish kernel
    if (focal_wide() && is_bf16 && !(V & 1) && !(ld & 1) && !(reinterpret_cast<uintptr_t>(logits) & 3))
        hipLaunchKernelGGL(focal_fwd_pairs_kernel, dim3((unsigned)rows), dim3(FC_WIDE), 0, st, (const unsigned*)logits,
                           labels, row_ce, row_lse, loss, count, (int)V, ld, gamma);
    else if (is_bf16)
        hipLaunchKernelGGL((focal_fwd_kernel<true>), dim3((unsigned)rows), dim3(FC_THREADS), 0, st, logits, labels,
                           row_ce, row_lse, loss, count, (int)V, ld, gamma);
    else
        hipLaunchKernelGGL((focal_fwd_kernel<false>), dim3((unsigned)rows), dim3(FC_THREADS), 0, st, logits, labels,
                           row_ce, row_lse, loss, count, (int)V, ld, gamma);
    int rc = clv_check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(focal_finish_kernel, dim3(1), dim3(1), 0, st, loss, (const float*)count);
    return clv_check_launch();
}

static int focal_bwd_impl(const void* logits, int32_t is_bf16, const int64_t* labels, const float* row_ce,
                          const float* row_lse, const float* count, const float* dloss, void* dlogits, int64_t rows, int32_t V,
                          int64_t ld, float gamma, void* stream) {
    if (!logits || !labels || !row_ce || !row_lse || !count || !dloss || !dlogits || rows <= 0 || V <= 0 || ld < V)
        return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (focal_wide() && is_bf16 && !(V & 1) && !(ld & 1) &&
        !((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 3))
        hipLaunchKernelGGL(focal_bwd_pairs_kernel, dim3((unsigned)rows), dim3(FC_WIDE), 0, st, (const unsigned*)logits,
                           labels, row_ce, row_lse, count, dloss, (unsigned*)dlogits, (int)V, ld, gamma);
    else if (is_bf16)
        hipLaunchKernelGGL((focal_bwd_kernel<true>), dim3((unsigned)rows), dim3(FC_THREADS), 0, st, logits, labels,
                           row_ce, row_lse, count, dloss, dlogits, (int)V, ld, gamma);
    else
        hipLaunchKernelGGL((focal_bwd_kernel<false>), dim3((unsigned)rows), dim3(FC_THREADS), 0, st, logits, labels,
                           row_ce, row_lse, count, dloss, dlogits, (int)V, ld, gamma);
    return clv_check_launch();
}

extern "C" int clv_focal_ce_fwd(const void* logits, int32_t is_bf16, const int64_t* labels, float* row_ce,
                                float* row_lse, float* loss, float* count, int64_t rows, int32_t V, float gamma,
                                void* stream) {
    return focal_fwd_impl(logits, is_bf16, labels, row_ce, row_lse, loss, count, rows, V, V, gamma, stream);
}

extern "C" int clv_focal_ce_bwd(const void* logits, int32_t is_bf16, const int64_t* labels, const float* row_ce,
                                const float* row_lse, const float* count, const float* dloss, void* dlogits,
                                int64_t rows, int32_t V, float gamma, void* stream) {
    return focal_bwd_impl(logits, is_bf16, labels, row_ce, row_lse, count, dloss, dlogits, rows, V, V, gamma, stream);
}

// The same on rows of stride ld >= V elements (logits and dlogits alike): the padded [rows, ld] score buffer the MLM decoder
// GEMM writes (V = 30522 is not a multiple of 8; ld = 30528 is).  The backward ZEROES the padding columns [V, ld) of dlogits,
// so that the decoder's input- and weight-gradient GEMMs can contract over ld.
extern "C" int clv_focal_ce_fwd_ld(const void* logits, int32_t is_bf16, const int64_t* labels, float* row_ce,
                                   float* row_lse, float* loss, float* count, int64_t rows, int32_t V, int64_t ld,
                                   float gamma, void* stream) {
    return focal_fwd_impl(logits, is_bf16, labels, row_ce, row_lse, loss, count, rows, V, ld, gamma, stream);
}

extern "C" int clv_focal_ce_bwd_ld(const void* logits, int32_t is_bf16, const int64_t* labels, const float* row_ce,
                                   const float* row_lse, const float* count, const float* dloss, void* dlogits,
                                   int64_t rows, int32_t V, int64_t ld, float gamma, void* stream) {
    return focal_bwd_impl(logits, is_bf16, labels, row_ce, row_lse, count, dloss, dlogits, rows, V, ld, gamma, stream);
}

extern "C" int64_t clv_infonce_work_floats(int32_t G, int32_t Dm) { return nce_work_floats(G, Dm); }

// one thread per exclusive row / column (6 G of them): whole waves, at most the kernel's 1024 (a per-rank batch of 8 needs one)
static int nce_loss_threads(int G) {
    const int t = (6 * G + 63) / 64 * 64;
    return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}

static int nce_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                    int batch, int64_t sA, int64_t sB, int64_t sC, float alpha, hipStream_t st, int outer = 1,
                    int64_t s2 = 0) {
    const dim3 grid((N + 15) / 16, (M + 15) / 16, batch * outer);
    if (K >= 256 && (int64_t)grid.x * grid.y * batch <= 512)       // few tiles (per evaluation), long contraction: 8 waves share a tile
        hipLaunchKernelGGL(sgemm_nt_kernel<8>, grid, dim3(512), 0, st, A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha,
                           batch, s2);
    else
        hipLaunchKernelGGL(sgemm_nt_kernel<1>, grid, dim3(64), 0, st, A, B, C, M, N, K, lda, ldb, ldc, sA, sB, sC, alpha,
                           batch, s2);
    return clv_check_launch();
}

extern "C" int clv_infonce_fwd(const float* e0, const float* e1, const float* e2, const float* e3, float* out,
                               float* work, int32_t G, int32_t Dm, int32_t ld, float temperature, float margin,
                               void* stream) {
    if (!e0 || !e1 || !e2 || !e3 || !out || !work || G <= 0 || Dm <= 0 || ld < Dm || temperature <= 0.f)
        return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NceWork W = nce_carve(work, G, Dm);
    hipLaunchKernelGGL(nce_normalize_kernel, dim3((4 * G + 3) / 4), dim3(256), 0, st, e0, e1, e2, e3, W, (int)G, (int)Dm,
                       (int)ld);
    int rc = clv_check_launch();
    if (rc) return rc;
    // sim[k] = en0 . en_{k+1}^T / temperature
    rc = nce_gemm(W.en, W.en + (int64_t)G * Dm, W.sim, G, G, Dm, Dm, Dm, G, 3, 0, (int64_t)G * Dm, (int64_t)G * G,
                  1.0f / temperature, st);
    if (rc) return rc;
    hipLaunchKernelGGL(nce_loss_kernel, dim3(1), dim3(nce_loss_threads(G)), 0, st, W, out, (int)G, margin);
    return clv_check_launch();
}

extern "C" int clv_infonce_bwd(const float* e0, const float* e1, const float* e2, const float* e3, const float* dout,
                               const float* work, float* d0, float* d1, float* d2, float* d3, int32_t G, int32_t Dm,
                               int32_t ldd, float temperature, float margin, void* stream) {
    if (!dout || !work || !d0 || !d1 || !d2 || !d3 || G <= 0 || Dm <= 0 || ldd < Dm) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NceWork W = nce_carve(const_cast<float*>(work), G, Dm);
    int64_t total = (int64_t)3 * G * G;
    int grid = (int)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(nce_dsim_kernel, dim3(grid), dim3(256), 0, st, W, dout, (int)G, margin);
    int rc = clv_check_launch();
    if (rc) return rc;
    const float it = 1.0f / temperature;
    // d en0[i][d] = sum_{k,j} dsim[i][(k,j)] * en_{k+1}[j][d] / T :  A = dsim [G][3G], B = enT[d][(k+1, j)]
    rc = nce_gemm(W.dsim, W.enT + G, W.den, G, Dm, 3 * G, 3 * G, 4 * G, Dm, 1, 0, 0, 0, it, st);
    if (rc) return rc;
    // d en_{k+1}[j][d] = sum_i dsimT[k][j][i] * en0[i][d] / T   :  A = dsimT[k] [G][G], B = enT[d][(0, i)]
    rc = nce_gemm(W.dsimT, W.enT, W.den + (int64_t)G * Dm, G, Dm, G, G, 4 * G, Dm, 3, (int64_t)G * G, 0,
                  (int64_t)G * Dm, it, st);
    if (rc) return rc;
    hipLaunchKernelGGL(nce_norm_bwd_kernel, dim3((4 * G + 3) / 4), dim3(256), 0, st, W, d0, d1, d2, d3, (int)G, (int)Dm,
                       (int)ldd);
    return clv_check_launch();
}

static bool nce_pair_setup(NcePair& P, const float* packed, float* work, const int32_t* slots, int G, int k, int Dm) {
    const int64_t ws = nce_work_floats(G, Dm);
    for (int d = 0; d < 2; ++d) {
        P.w[d] = nce_carve(work + d * ws, G, Dm);
        for (int q = 0; q < 4; ++q) {
            const int sl = slots[d * 4 + q];
            if (sl < 0 || sl >= k) return false;
            for (int r = 0; r < q; ++r)
                if (slots[d * 4 + r] == sl) return false;               // the four slots of a direction are distinct
            P.slot[d][q] = sl;
            P.e[d][q] = packed ? packed + (int64_t)sl * Dm : nullptr;
        }
    }
    for (int q = 0; q < 4; ++q) P.dout[q] = nullptr;
    return true;
}

extern "C" int clv_infonce_pair_fwd(const float* packed, const int32_t* slots, float* out, float* work, int32_t G,
                                    int32_t k, int32_t Dm, float temperature, float margin, void* stream) {
    if (!packed || !slots || !out || !work || G <= 0 || k < 4 || Dm <= 0 || temperature <= 0.f) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NcePair P;
    if (!nce_pair_setup(P, packed, work, slots, G, k, Dm)) return CLV_ERR_ARG;
    const int64_t ws = nce_work_floats(G, Dm);
    hipLaunchKernelGGL(nce_pair_normalize_kernel, dim3((4 * G + 3) / 4, 2), dim3(256), 0, st, P, (int)G, (int)Dm,
                       (int)(k * Dm));
    int rc = clv_check_launch();
    if (rc) return rc;
    rc = nce_gemm(P.w[0].en, P.w[0].en + (int64_t)G * Dm, P.w[0].sim, G, G, Dm, Dm, Dm, G, 3, 0, (int64_t)G * Dm,
                  (int64_t)G * G, 1.0f / temperature, st, 2, ws);
    if (rc) return rc;
    hipLaunchKernelGGL(nce_pair_loss_kernel, dim3(2), dim3(nce_loss_threads(G)), 0, st, P, out, (int)G, margin);
    return clv_check_launch();
}

extern "C" int clv_infonce_pair_bwd(const float* const* dout, const float* work, const int32_t* slots, float* dpacked,
                                    int32_t G, int32_t k, int32_t Dm, float temperature, float margin, void* stream) {
    if (!dout || !work || !slots || !dpacked || G <= 0 || k < 4 || Dm <= 0 || temperature <= 0.f) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NcePair P;
    if (!nce_pair_setup(P, nullptr, const_cast<float*>(work), slots, G, k, Dm)) return CLV_ERR_ARG;
    for (int q = 0; q < 4; ++q) P.dout[q] = dout[q];
    const int64_t ws = nce_work_floats(G, Dm);
    int64_t total = (int64_t)3 * G * G;
    int grid = (int)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(nce_pair_dsim_kernel, dim3(grid, 2), dim3(256), 0, st, P, (int)G, margin);
    int rc = clv_check_launch();
    if (rc) return rc;
    const float it = 1.0f / temperature;
    const NceWork& W = P.w[0];
    rc = nce_gemm(W.dsim, W.enT + G, W.den, G, Dm, 3 * G, 3 * G, 4 * G, Dm, 1, 0, 0, 0, it, st, 2, ws);
    if (rc) return rc;
    rc = nce_gemm(W.dsimT, W.enT, W.den + (int64_t)G * Dm, G, Dm, G, G, 4 * G, Dm, 3, (int64_t)G * G, 0,
                  (int64_t)G * Dm, it, st, 2, ws);
    if (rc) return rc;
    hipLaunchKernelGGL(nce_pair_norm_bwd_kernel, dim3((k * G + 3) / 4), dim3(256), 0, st, P, dpacked, (int)G, (int)k,
                       (int)Dm);
    return clv_check_launch();
}

extern "C" int64_t clv_normsoftmax_work_floats(int32_t G, int32_t Dm) { return ns_work_floats(G, Dm > 0 ? Dm : 1); }

extern "C" int clv_normsoftmax_fwd(const float* video, const float* text, const float* sim_mat, float* out,
                                   float* work, int32_t G, int32_t Dm, float temperature, float eps, void* stream) {
    if (!out || !work || G <= 0) return CLV_ERR_ARG;
    if (!sim_mat && (!video || !text || Dm <= 0 || temperature <= 0.f || eps <= 0.f)) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NsWork W = ns_carve(work, G, Dm > 0 ? Dm : 1);
    if (!sim_mat) {
        hipLaunchKernelGGL(ns_normalize_kernel, dim3((2 * G + 3) / 4), dim3(256), 0, st, video, text, W, (int)G, (int)Dm,
                           eps);
        int rc = clv_check_launch();
        if (rc) return rc;
        rc = nce_gemm(W.en, W.en + (int64_t)G * Dm, W.sim, G, G, Dm, Dm, Dm, G, 1, 0, 0, 0, 1.0f / temperature, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(ns_loss_kernel, dim3(1), dim3(1024), 0, st, W, sim_mat ? sim_mat : (const float*)W.sim, out,
                       (int)G);
    return clv_check_launch();
}

extern "C" int clv_normsoftmax_bwd(const float* sim_mat, const float* dout, const float* work, float* dvideo,
                                   float* dtext, float* dsim, int32_t G, int32_t Dm, float temperature,
                                   void* stream) {
    if (!dout || !work || G <= 0) return CLV_ERR_ARG;
    if (sim_mat ? !dsim : (!dvideo || !dtext || Dm <= 0 || temperature <= 0.f)) return CLV_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    NsWork W = ns_carve(const_cast<float*>(work), G, Dm > 0 ? Dm : 1);
    const int64_t total = (int64_t)G * G;
    int grid = (int)((total + 255) / 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(ns_dsim_kernel, dim3(grid), dim3(256), 0, st, W, sim_mat ? sim_mat : (const float*)W.sim, dout,
                       sim_mat ? dsim : W.dsim, (int)G);
    int rc = clv_check_launch();
    if (rc || sim_mat) return rc;
    const float it = 1.0f / temperature;
    // d en_v[i][d] = sum_j dsim[i][j] en_t[j][d] / t ;  d en_t[j][d] = sum_i dsim[i][j] en_v[i][d] / t
    rc = nce_gemm(W.dsim, W.enT + G, W.den, G, Dm, G, G, 2 * G, Dm, 1, 0, 0, 0, it, st);
    if (rc) return rc;
    rc = nce_gemm(W.dsimT, W.enT, W.den + (int64_t)G * Dm, G, Dm, G, G, 2 * G, Dm, 1, 0, 0, 0, it, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ns_norm_bwd_kernel, dim3((2 * G + 3) / 4), dim3(256), 0, st, W, dvideo, dtext, (int)G, (int)Dm);
    return clv_check_launch();
}
