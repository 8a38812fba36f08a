/* clover_hip.h — C ABI of libclover_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the Clover video-text pre-training hot path.  The
 * reference (LeeYN-43/Clover) is 100 % Python over torch/cuDNN/cuBLAS and has no
 * native FFI of its own; each entry point below replaces the torch op group that
 * the cited reference lines execute (paths relative to /root/reference/).  A
 * reference maintainer binds them with ctypes from the registered nn.Modules —
 * see INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers + sizes; no torch / C++ types cross the ABI;
 *   - every pointer is DEVICE memory unless its name ends in _host;
 *   - bf16 tensors are `void*` (uint16 storage), row-major, innermost contiguous;
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous on it,
 *     allocate nothing, and are capturable into a hipGraph;
 *   - return 0 on success, negative CLV_ERR_* otherwise; no exceptions.
 */
#ifndef CLOVER_HIP_H
#define CLOVER_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLV_ABI_VERSION 17
#define CLV_ERR_ARG (-1)
#define CLV_ERR_UNSUPPORTED (-2)
#define CLV_ERR_LAUNCH (-3)

int clv_abi_version(void);
/* The 16-bit element type this build of the library computes in: 0 = bf16 (libclover_hip.so), 1 = IEEE fp16
 * (libclover_hip_f16.so: the same sources compiled with -DCLV_HALF_F16 — the reference's own arithmetic type,
 * configs/exp_local/pretrain_webvid_cc3m.py:21, mmaction/core/hooks/fp16_utils.py:215-259).  Every `bf16` in the comments
 * below reads "the library's 16-bit element type". */
int clv_half_type(void);

/* ------------------------------------------------------------------ attention
 * One kernel family serves both attention flavours of the path:
 *   mode 1: 3-D shifted-window MHA — WindowAttention3D.forward
 *           (mmaction/models/backbones/swin_transformer_3d.py:375-400) fused with
 *           the cyclic roll + window_partition before it (:459-466) and the
 *           window_reverse + un-roll after it (:470-476): q/k/v are read from, and
 *           o is written to, the natural [B,D,H,W,*] token layout;
 *   mode 0: full self-attention over [video ‖ text] / text with an additive key
 *           mask — transformers 4.6.1 BertSelfAttention as called from
 *           bert_from_hugface.py:30 and cross_transformer.py:109-110.
 * softmax( q·kᵀ·scale + bias[h][i][j] + (rid[i]!=rid[j] ? -100 : 0) + kmask[b][j] ) · v
 *
 * Relative-position bias (mode 1): the kernels take the TABLE, not a gathered [nH][N][N] tensor:
 *   bias[h][i][j] = table[ lin(i) - lin(j) + c ][h],   lin(n) = (n / (bwh*bww)) * (2bwh-1)(2bww-1)
 *                                                              + ((n / bww) % bwh) * (2bww-1) + n % bww
 * which is relative_position_bias_table[relative_position_index[:N,:N]] of :343-360,386 for the window
 * (bwd,bwh,bww) the table was built for (the module's full window_size; N may be a clipped window's
 * token count).  The head's table lives in LDS (2535 entries for (8,7,7)); the backward sums the per-window dS
 * over the windows and scatters the sums into d table — no gathered [nH][N][N] bias or bias gradient exists.
 */
typedef struct ClvAttnGeom {
    int32_t mode;           /* 0 sequence, 1 shifted 3-D window */
    int32_t groups;         /* B (mode 0) or B*nW (mode 1) */
    int32_t N;              /* tokens per group: S or wd*wh*ww */
    int32_t nH;             /* heads */
    int32_t hd;             /* head dim: 16, 32 or 64 */
    int32_t D, H, W;        /* mode 1: (padded) feature dims */
    int32_t wd, wh, ww;     /* mode 1: effective window (get_window_size, :302-315) */
    int32_t sd, sh, sw;     /* mode 1: effective shift */
    int32_t ldq, ldk, ldv, ldo; /* row strides (elements) of q,k,v and o/do/dq.. */
    int32_t bwd, bwh, bww;  /* window the bias table was built for: the table has (2bwd-1)(2bwh-1)(2bww-1) rows; 0,0,0 = none */
    float scale;            /* head_dim^-0.5 */
    float dropout_p;        /* dropout on the attention probabilities (HF attention_probs_dropout_prob), 0 = off */
    const int32_t* dbias_index; /* backward, optional: the device table clv_attn_dbias_index() built for this window
                                   geometry (N, bwd, bwh, bww) — the table-gradient gather then does no index arithmetic */
    void* work;                 /* mode 0 with 448 < N <= clv_attn_seq_max_keys() (the 32-frame fusion sequence, 816 tokens,
                                   cross_transformer.py:89-110): clv_attn_seq_work_bytes() of scratch.  K / V (Q / dO) of one
                                   (sample, head) then exceed the LDS, so the staged tokens are split in two parts, each
                                   with its own workgroups; the parts' results (o + lse; dq; dk / dv) meet in a merge
                                   kernel.
                                   mode 1, backward with a bias table, optional: clv_attn_dbias_partial_bytes() of 16-byte
                                   aligned memory that receives the slices' fp32 partial sums of the table gradient INSTEAD
                                   of the tail of clv_attn_bwd's `work` — a caller that gathers later (stage bit 8 +
                                   clv_attn_dbias_gather_batch) keeps only this and releases the dS scratch (GBs at 32
                                   frames).  Not read otherwise. */
} ClvAttnGeom;

/* lse: float [groups][nH][N].  bias: the module's relative_position_bias_table, float [rows][nH], or NULL.  rid: int32 [nW][N] region
 * ids of compute_mask (:548-562) or NULL.  kmask: float [groups][N] additive or NULL.
 * seed: device uint64[1], read when dropout_p > 0; the mask is a counter-based hash of
 * (seed, group, head, query, key), so the backward regenerates it from the same seed. */
int64_t clv_attn_seq_work_bytes(const ClvAttnGeom* geom);
int clv_attn_seq_max_keys(void);
int clv_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                 const float* bias, const int32_t* rid, const float* kmask, const void* seed,
                 const ClvAttnGeom* geom_host, void* stream);

/* dq/dk/dv use strides ldq/ldk/ldv; dout uses ldo.  dbias (float [rows][nH], the gradient of the table)
 * is ACCUMULATED into (caller zeroes).  dsum: float scratch [groups][nH][N].
 * work: scratch of clv_attn_bwd_work_bytes() bytes (the bf16 dS of every (group, head), summed over the groups
 * and scattered into dbias by a second kernel), required iff bias != NULL. */
int64_t clv_attn_bwd_work_bytes(const ClvAttnGeom* geom_host);
/* 1 if clv_attn_bwd runs this geometry (with a bias table) as ONE kernel for dQ / dK / dV (round 4: window mode, head dim 32,
 * 196- / 392-token windows) — `stages` then knows the masks 5 (that kernel) and 2 (the table gradient) instead of 1 / 2 / 4. */
int clv_attn_bwd_one_kernel(const ClvAttnGeom* geom_host);
/* int32 [clv_attn_dbias_index_count()]: fragment offset of the (query, key) pair of every (bias-table row, key), -1 = none;
 * depends on (N, bwd, bwh, bww) only — build once, pass in ClvAttnGeom.dbias_index of every backward call. */
int64_t clv_attn_dbias_index_count(const ClvAttnGeom* geom_host);
int clv_attn_dbias_index(const ClvAttnGeom* geom_host, int32_t* out, void* stream);
int clv_attn_bwd(const void* q, const void* k, const void* v, const void* o, const void* dout,
                 const float* lse, const float* bias, const int32_t* rid, const float* kmask,
                 void* dq, void* dk, void* dv, float* dbias, float* dsum, void* work,
                 const void* seed, int32_t stages, const ClvAttnGeom* geom_host, void* stream);
/* Deferred table gradient (swin_transformer_3d.py:382-386, the backward of the relative_position_bias_table gather): with
 * bit 8 set in `stages` (e.g. 15 = everything but the gather) clv_attn_bwd leaves the slices' partial sums of dS in `work` and
 * does NOT add them into dbias; clv_attn_dbias_gather_entry describes that pending gather (work / dbias / geom as passed to
 * clv_attn_bwd, geom->dbias_index required) and clv_attn_dbias_gather_batch runs up to CLV_DBIAS_GATHER_MAX of them as ONE
 * launch (dbias is accumulated into).  `work` must stay untouched in between.  Nothing reads a table gradient before the
 * optimizer, so a backward pass gathers all its attention blocks at its end. */
#define CLV_DBIAS_GATHER_MAX 32
typedef struct ClvDbiasGather {
    const void* partial;       /* float partial sums inside the work buffer (16-byte aligned; the batch call CONSUMES them:
                                  it adds the slices into slice 0 before it gathers) */
    void* dtable;              /* float [rows][nH] */
    const void* index;         /* int32 table of clv_attn_dbias_index() */
    int64_t split_stride;
    int32_t nkt, nH, N, nsplit, slot0, nslots, block_begin, pad;    /* block_begin, pad: set by the batch call (block range;
                                                                       heads per wave) — whatever the caller leaves there */
} ClvDbiasGather;
int clv_attn_dbias_gather_entry(const ClvAttnGeom* geom_host, void* work, float* dbias, ClvDbiasGather* out);
int64_t clv_attn_dbias_partial_bytes(const ClvAttnGeom* geom_host);    /* size of ClvAttnGeom.work in mode 1 (see there) */
int clv_attn_dbias_gather_batch(const ClvDbiasGather* entries, int32_t n, void* stream);

/* ---- unfused attention for long sequences (N > 448 keys: K/V of one (group, head) no longer fit LDS in the
 * kernels above; the fusion encoder at 32 frames has 16*49 + 32 = 816 tokens).  Q.K^T, P.V and their gradients are
 * batched library GEMMs; these two kernels are what sits between them (HF BertSelfAttention, transformers 4.6.1
 * modeling_bert.py, reached through cross_transformer.py:95-108): one wave per row.
 * fwd: p[r][j] = softmax_j(scores[r][j]*scale + kmask[r / rows_per_group][j]); pd = p * keep/(1-p_drop) with the
 *      same counter-based mask as clv_attn_* (row id r, key j).  scores/p/pd bf16 [rows][ld], S <= 2048 keys;
 *      kmask float [groups][S] or NULL; pd / seed may be NULL when dropout_p == 0.
 * bwd: ds = p * (dp - sum_j p_j dp_j) * scale with dp = dpd * keep/(1-p_drop); ds may alias dpd. */
int clv_softmax_rows_fwd(const void* scores, const float* kmask, void* p, void* pd, const void* seed, int64_t rows,
                         int32_t S, int32_t ld, int32_t rows_per_group, float scale, float dropout_p, void* stream);
int clv_softmax_rows_bwd(const void* p, const void* dpd, void* ds, const void* seed, int64_t rows, int32_t S,
                         int32_t ld, float scale, float dropout_p, void* stream);
/* stages: 0 = everything; otherwise a bit mask 1 = dQ (+ dS scratch, dsum) kernel, 2 = dS -> table-gradient
 * reduction, 4 = dK/dV kernel — lets a profiler bracket each kernel of the call with its own events. */

/* ------------------------------------------------------------------ LayerNorm
 * nn.LayerNorm over the last dim (every norm site: swin_transformer_3d.py:450,483,
 * 541,685,238; HF BERT LayerNorm eps 1e-12; cross_transformer.py:97).
 * x,y [rows][C] bf16 (is_f32 = 0) or float (is_f32 = 1: the fp32 projection heads and
 * losses, ssl_head.py:50-56, contrastive_loss.py:102); gamma,beta float [C]; mean,rstd
 * float [rows].  res (same type as x, may be NULL): y = LN(x + res) (BERT post-LN residual;
 * the Swin residual adds of :498,503 fused into the following norm).  sum_out (may be NULL)
 * receives x + res — the updated residual stream. */
/* Optional operand transforms / extra operands (pass NULL for a plain LayerNorm; C % 8 == 0 and C <= 3072 only):
 *   forward   t = keep(x) * x / (1 - drop_p) * xscale[row / rows_per_sample] + res,  y = LN(t)
 * i.e. the nn.Dropout on the sub-layer output (BertSelfOutput / BertOutput: LayerNorm(dropout(dense(h)) + input))
 * and the per-sample DropPath factor (swin_transformer_3d.py:498,503) applied while the row is loaded.  The
 * dropout mask is a pure function of (*seed, row, column); the backward regenerates it.
 *   backward  d t = LN-backward(dy + dy2) + dsum;  dres (may be NULL) receives d t, dx receives d t times the
 * multiplier x carried.  x_is_sum = 1: the x argument of the backward already holds t (a saved sum_out). */
typedef struct ClvLnExtra {
    const float* xscale;       /* per-sample factor on x, or NULL */
    int32_t rows_per_sample;
    float drop_p;              /* dropout probability on x, 0 = off */
    const void* seed;          /* device uint64, required when drop_p > 0 */
    const void* dy2;           /* backward: second upstream gradient (same type as dy), or NULL */
    void* dres;                /* backward: gradient wrt res when it differs from dx, or NULL */
    int32_t x_is_sum;
    /* PatchMerging (swin_transformer_3d.py:531-539) folded into the LayerNorm that follows it: with gather_c > 0 the
     * normalised row of width C = 4*gather_c is [x(2h,2w) | x(2h+1,2w) | x(2h,2w+1) | x(2h+1,2w+1)] of the UN-gathered
     * x / res [.., 2*gather_h2, 2*gather_w2, gather_c]; rows = .. * gather_h2 * gather_w2; y stays [rows][C]; the
     * backward writes dx / dres in the un-gathered layout (every element exactly once).  No sum_out / dsum / dropout. */
    int32_t gather_c, gather_h2, gather_w2;
    /* backward: leave the per-block partials in `partial` and skip the reduction launch — the caller folds the
     * partials of many LayerNorms into their dgamma / dbeta with ONE clv_ln_reduce_batch launch (needs_reduce tells
     * whether this launch wrote partials at all: up to 256 blocks add into dgamma / dbeta directly) */
    int32_t no_reduce;
    /* forward, fp8 path (vector kernels: C % 8 == 0 and one of their widths): also emit y as OCP e4m3 with one scale per row
     * — q8 [rows][C] bytes = y / qscale[row], qscale[row] = max |y[row]| / 448 — the activation operand of clv_gemm_nt_fp8,
     * produced here instead of by a clv_quant_fp8_rows pass over y.  Both NULL: off. */
    void* q8;
    float* qscale;
} ClvLnExtra;
int clv_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta,
                      void* y, void* sum_out, float* mean, float* rstd, int64_t rows, int32_t C,
                      float eps, int32_t is_f32, const ClvLnExtra* extra, void* stream);
/* dx [rows][C] (gradient wrt x and, identically, wrt res unless extra says otherwise) = LN backward of dy
 * (+ dsum, the gradient arriving on sum_out, may be NULL); dgamma,dbeta float [C]
 * ACCUMULATED into (caller zeroes); partial: float scratch [2][nblk][C] with nblk = clv_layernorm_bwd_blocks(). */
int clv_layernorm_bwd_blocks(int64_t rows, int32_t C);
int clv_layernorm_bwd(const void* dy, const void* x, const void* res, const float* gamma,
                      const float* mean, const float* rstd, const void* dsum, void* dx, float* dgamma, float* dbeta,
                      float* partial, int64_t rows, int32_t C, int32_t is_f32, const ClvLnExtra* extra, void* stream);

/* Batched reduction of LayerNorm-backward partials (launches made with extra->no_reduce): dgamma[c] += sum_i
 * partial[0][i][c], dbeta[c] += sum_i partial[1][i][c] for up to CLV_LN_REDUCE_MAX entries (HOST array) in one launch. */
#define CLV_LN_REDUCE_MAX 64
typedef struct ClvLnReduceEntry {
    const float* partial;      /* [2][nblk][C] */
    float* dgamma;
    float* dbeta;
    int32_t nblk, C, block_begin, pad;
} ClvLnReduceEntry;
int clv_layernorm_bwd_needs_reduce(int64_t rows, int32_t C);
int clv_ln_reduce_batch(const ClvLnReduceEntry* entries, int32_t n, void* stream);

/* LayerNorm affine folded into the following Linear (the fused LN + projection kernels, clv_rowgemm with
 * standardise = 1, consume standardised rows):  wf[n][k] = bf16(w[n][k] * gamma[k]),  bf[n] = b[n] + sum_k w[n][k] beta[k]
 * (w float [N][K], b float [N] or NULL).  Backward: given d wf (float [N][K]) and d bf (float [N]),
 * dw / db / dgamma / dbeta are ACCUMULATED into. */
int clv_ln_fold_fwd(const float* w, const float* b, const float* gamma, const float* beta, void* wf, float* bf,
                    int32_t N, int32_t K, void* stream);
int clv_ln_fold_bwd(const float* dwf, const float* dbf, const float* w, const float* gamma, const float* beta,
                    float* dw, float* db, float* dgamma, float* dbeta, int32_t N, int32_t K, void* stream);

/* ------------------------------------------------------------------ BatchNorm1d of the projection heads
 * nn.BatchNorm1d on [B][D] fp32 rows — the ln=False / text_bn=True variants of NCEHeadForMM / NCEHeadForVision /
 * NCEHeadForText (mmaction/models/heads/ssl_head.py:50-66,175-186,252-262).  training != 0: batch statistics (biased
 * variance), running_mean / running_var (may both be NULL) updated with `momentum` and the unbiased variance; training == 0:
 * the running statistics normalise.  save_mean / save_rstd (float [D]) are outputs the backward reads.  Backward: dx (or NULL),
 * dgamma, dbeta (or NULL) are WRITTEN; in training mode dx carries the dependence of the statistics on the batch. */
int clv_batchnorm1d_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float* y, float* save_mean, float* save_rstd, int32_t B, int32_t D, float eps, float momentum,
                        int32_t training, void* stream);
int clv_batchnorm1d_bwd(const float* dy, const float* x, const float* gamma, const float* save_mean, const float* save_rstd,
                        float* dx, float* dgamma, float* dbeta, int32_t B, int32_t D, int32_t training, void* stream);

/* ------------------------------------------------------------------ GELU (erf)
 * nn.GELU / HF 'gelu' (swin_transformer_3d.py:264; BertIntermediate; ssl_head.py:53). */
int clv_gelu_fwd(const void* x, void* y, int64_t n, int32_t is_f32, void* stream);
int clv_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, int32_t is_f32, void* stream);

/* ------------------------------------------------------------------ patch embed
 * PatchEmbed3D.forward (swin_transformer_3d.py:671-688: conv3d k=s=(2,4,4) as a GEMM
 * over non-overlapping patches + bias + LayerNorm(C)) fused with the mask-token blend
 * of SwinTransformer3D.forward (:222-230).  x float [B][3][T][H][W] (T even, H,W
 * multiples of 4 — the caller pads, :675-680); w bf16 [C][96] (= proj.weight.flatten(1));
 * out_clean / out_masked bf16 [B][T/2][H/4][W/4][C] channels-last; either may be NULL.
 * vmask int64 [B][mh][mw] (the v_token_mask) and mask_token float [C] — required iff
 * out_masked != NULL.  gamma/beta float [C] or both NULL (patch_norm=False).
 * z_out (bf16 [M][C] pre-norm conv output), mean, rstd (float [M]): saved for backward,
 * may be NULL.  C in {48, 96, 128}. */
int clv_patch_embed_fwd(const float* x, const void* w, const float* bias, const float* gamma,
                        const float* beta, const float* mask_token, const int64_t* vmask,
                        void* out_clean, void* out_masked, void* z_out, float* mean, float* rstd,
                        int32_t B, int32_t T, int32_t H, int32_t W, int32_t C, int32_t mh,
                        int32_t mw, float eps, void* stream);
/* Backward of the mask-token blend (:222-230): dy = d_clean + d_masked (1 - w) (bf16 [M][C]; either input may be NULL)
 * and dmask_token float [C] += sum over tokens of d_masked w (ACCUMULATED; caller zeroes), w from vmask as above. */
int clv_patch_embed_blend_bwd(const void* dclean, const void* dmasked, const int64_t* vmask, void* dy,
                              float* dmask_token, int32_t B, int32_t T, int32_t H, int32_t W, int32_t C,
                              int32_t mh, int32_t mw, void* stream);
/* im2col of the clip into bf16 patches [M][96] (k = c*32 + dt*16 + dy*4 + dx): the operand
 * of the weight-gradient GEMM dW = dZ^T * patches (the conv3d weight grad of :681). */
int clv_im2col_patches(const float* x, void* patches, int32_t B, int32_t T, int32_t H, int32_t W,
                       void* stream);

/* ------------------------------------------------------------------ Linear weight/bias gradient
 * dW[n][k] += sum_m dY[m][n] X[m][k], db[n] += sum_m dY[m][n] — the weight gradient of every
 * token-parallel nn.Linear on the path (qkv/proj/fc1/fc2/reduction, swin_transformer_3d.py:257-259,
 * 361-363,518; the conv3d weight of :665 through clv_im2col_patches) for huge M and small N x K,
 * where the contraction (token) dimension is split over the whole chip.  dy bf16 [M][N] (row stride
 * ldy), x bf16 [M][K] (row stride ldx); dw float [N][K], db float [N] or NULL — both ACCUMULATED
 * into.  N, K, ldy, ldx multiples of 8.  work: float scratch, clv_linear_wgrad_work_floats().
 * xmean / xrstd (float [M], both or neither): X rows are standardised on load,
 * x_hat = (x - mean) * rstd — the LayerNorm output the fused clv_rowgemm forward never stored. */
int64_t clv_linear_wgrad_work_floats(int64_t M, int32_t N, int32_t K);
int clv_linear_wgrad(const void* dy, const void* x, float* dw, float* db, float* work, int64_t M,
                     int32_t N, int32_t K, int32_t ldy, int32_t ldx, const float* xmean,
                     const float* xrstd, int32_t stages, void* stream);
/* stages: 0 = both kernels; bit 1 = split-M partial kernel, bit 2 = fold of the partials (profiling, or a deferred
 * fold: a backward pass may run its weight-gradient launches with stages = 1 and fold all their partials at once).
 * clv_linear_wgrad_splits: the number of M-slices (= fp32 partial slabs [N*K dW | N db] in work) the launch uses; a fold
 * is needed when it is > 1 or xmean is given.  clv_wgrad_fold_batch: dw[e] += sum_s partial[s][e], db likewise, for up
 * to CLV_FOLD_MAX launches in one kernel (entries is a HOST array; sg_shift / block_begin are filled in by the callee). */
#define CLV_FOLD_MAX 64
typedef struct ClvFoldEntry {
    const void* partial;   /* work of that clv_linear_wgrad call */
    void* dw;              /* float [N][K], accumulated into */
    void* db;              /* float [N] or NULL */
    int64_t nk, e2;        /* N*K and N*K + N */
    int32_t splits, sg_shift, block_begin;
    int32_t overwrite;     /* bit 0: dw = the sum, bit 1: db = the sum (an uninitialised temporary, or the step's first
                              gradient of a weight the engine does not clear), instead of += */
} ClvFoldEntry;
/* Grouped launch: the weight gradients of up to 80 Linear layers (a whole backward segment: nothing reads a weight
 * gradient before the optimizer) as ONE grid.  clv_linear_wgrad_batch_plan fills splits and work_floats of every entry
 * (fewer M-slices per problem than a stand-alone launch needs); the caller points `work` at work_floats floats,
 * clv_linear_wgrad_batch writes the fp32 partials [splits][N*K dW | N db] there, and clv_wgrad_fold_batch (entries with
 * the same work / splits) adds them into dW / db.  entries are HOST arrays. */
typedef struct ClvWgradEntry {
    const void* dy;            /* bf16 [M][N], row stride ldy */
    const void* x;             /* bf16 [M][K], row stride ldx */
    void* work;                /* float [work_floats] */
    float* dw;                 /* work_floats == 0 only (few rows, or a very large output: one slice, no partials, no fold): */
    float* db;                 /*   float [N][K] / [N] (or NULL) gradients, accumulated in place                        */
    int64_t M, work_floats;
    int32_t N, K, ldy, ldx, want_bias, splits;
    int32_t overwrite;         /* work_floats == 0 only; bit 0: dw = (stored, not added: its previous content is stale) */
    int32_t pad;
} ClvWgradEntry;
int clv_linear_wgrad_batch_plan(ClvWgradEntry* entries, int32_t n);
/* 1 if the plan gives this problem work_floats == 0 (one slice added into dw / db in place, without atomics): two such
 * problems with the SAME dw must not share a clv_linear_wgrad_batch call. */
int clv_linear_wgrad_in_place(int64_t M, int32_t N, int32_t K);
/* Tile class clv_linear_wgrad_batch gives the problem: 0 = wgrad_dma2_group_kernel (128 x 128 tiles), 1 =
 * wgrad_big_group_kernel<2, 2> (256 x 256; N and K multiples of 256, M > 1024) — the Linears of swin_transformer_3d.py:257-259,
 * 361-363,518; one launch per class present in the call.  (Round 5's opt-in class 4, shape-fitted (96 a) x (96 b) tiles,
 * moved 1.39 x the algorithmic bytes instead of 2.23 x but cost the step +0.15..0.25 ms and was removed in round 6:
 * profiles/r05_wgrad_tile_class.txt.) */
int clv_linear_wgrad_class(int64_t M, int32_t N, int32_t K);
int clv_linear_wgrad_batch(const ClvWgradEntry* entries, int32_t n, void* stream);
/* Gradient-norm partial sums from the kernels that write a weight gradient (round 6; the optimizer's norm pass of
 * mmcv_Fp16OptimizerHook.py:126-131 `clip_grads` without re-reading those gradients).  sumsq_slots: CLV_SUMSQ_SLOTS
 * accumulators, 16 floats (64 bytes) apart, zero before the backward.  An entry with overwrite bit 2 (value 4) set — in-place
 * entries of clv_linear_wgrad_batch_ss, any entry of clv_wgrad_fold_batch_ss — adds the sum of squares of the dW it STORES
 * (the final value: stored or accumulated) to a slot picked by its block index; clv_optim_prep_slots adds the slots to the
 * norm.  The caller's own clv_sumsq pass must then skip those tensors (clv_sumsq_ranges).  sumsq_slots = NULL: the plain
 * entry points. */
#define CLV_SUMSQ_SLOTS 64
int clv_linear_wgrad_batch_ss(const ClvWgradEntry* entries, int32_t n, float* sumsq_slots, void* stream);
int clv_wgrad_fold_batch_ss(const ClvFoldEntry* entries, int32_t n, float* sumsq_slots, void* stream);
int clv_linear_wgrad_splits(int64_t M, int32_t N, int32_t K);
int clv_wgrad_fold_batch(const ClvFoldEntry* entries, int32_t n, void* stream);

/* ------------------------------------------------------------------ token-parallel projections
 * Y[M][N] = epilogue( prologue(X)[M][K] * Wt[N][K]^T + bias ) — the QKV / proj / fc1 / fc2 Linears of
 * the high-resolution Swin stages and their input-gradient GEMMs (swin_transformer_3d.py:376,398,
 * 263-266), HBM-bound (K, N of a few hundred, M = 10^4..10^5 tokens), with the neighbouring
 * memory-bound ops folded in:
 *   prologue  standardise != 0: x = X (+ res; the sum is written to sum_out), then (x - mean) * rstd —
 *             nn.LayerNorm (:450,483) with its affine part folded into Wt/bias by the caller; mean, rstd
 *             (float [M]) are outputs; xhat_out (bf16 [M][K], row stride ldx, or NULL) receives the standardised rows —
 *             the operand of the weight gradient dW = dY^T x_hat;
 *   epilogue  0: + bias;  1: + bias, erf-GELU (:264), the pre-activation goes to pre_out;
 *             2: multiply by gelu'(pre_in) (GELU backward fused into the fc2 input-gradient GEMM).
 * X, res, sum_out bf16 [M][K] (row stride ldx); Wt bf16 [N][K] contiguous; bias float [N] or NULL;
 * Y, pre_in, pre_out bf16 [M][N] (row stride ldy).  K in {96,128,192,256,288,384,512,576,768}
 * (<= 256 with standardise), N % 8 == 0: query clv_rowgemm_supported(). */
int clv_rowgemm_supported(int32_t N, int32_t K, int32_t standardise);
int clv_rowgemm(const void* x, const void* res, void* sum_out, float* mean, float* rstd, void* xhat_out,
                const void* wt, const float* bias, const void* pre_in, void* y, void* pre_out, int64_t M, int32_t N,
                int32_t K, int32_t ldx, int32_t ldy, int32_t standardise, int32_t epilogue, float eps,
                void* stream);
/* The same with a per-sample factor on X inside the residual-add prologue: x = xscale[row / rows_per_sample] * X + res — the
 * DropPath factor of the attention branch whose residual add (swin_transformer_3d.py:498) the LayerNorm + fc1 kernel performs
 * (res required).  Round 5: the stage-0 block multiplied the branch by its factor with an elementwise pass each way. */
int clv_rowgemm_xs(const void* x, const void* res, void* sum_out, float* mean, float* rstd, void* xhat_out,
                   const void* wt, const float* bias, const void* pre_in, void* y, void* pre_out, int64_t M, int32_t N,
                   int32_t K, int32_t ldx, int32_t ldy, int32_t standardise, int32_t epilogue, float eps,
                   const float* xscale, int32_t rows_per_sample, void* stream);

/* ------------------------------------------------------------------ the MLP half of a stage-0 block as one kernel each way
 * (round 6) norm2 + Mlp + the residual add in front of them — swin_transformer_3d.py:482-483 (norm2, mlp), :262-268
 * (fc1, GELU, fc2), :498 / :503 (residual, DropPath) — for VideoSwin-T's stage-0 widths, C = 96 and hidden = 384
 * (clv_mlp_fused_supported): both weight matrices stay in LDS, the [M][hidden] activations never reach HBM.
 *   forward:  t = xscale[row / rows_per_sample] * a + res (written to sum_out: the new residual stream; res / xscale may be
 *             NULL), xhat = (t - mean) * rstd (mean, rstd float [M] are outputs, statistics of the fp32 t),
 *             out = GELU(xhat W1f^T + b1f) W2^T + b2.  a, res, sum_out, out bf16 [M][C] contiguous; w1f bf16 [hidden][C] and
 *             b1f float [hidden]: fc1 with the norm's affine part folded in (clv_ln_fold_fwd); w2 bf16 [C][hidden], b2 float
 *             [C] or NULL.
 *   backward: from tsum (= sum_out, or a when there was no res), mean, rstd, d out and the gradient d sum that arrives on the
 *             residual stream (bf16 [M][C] or NULL): recomputes xhat, the pre-activation, GELU and GELU'; d act = d out W2
 *             (w2t = W2^T, bf16 [hidden][C]); d pre = d act GELU'; d xhat = d pre W1f; d t = LayerNorm backward of d xhat
 *             (+ d sum).  Outputs: da = xscale * d t (gradient of a), dres = d t (gradient of res; NULL: not written — without
 *             xscale it equals da), and the operands of the weight-gradient GEMMs that follow (clv_linear_wgrad_batch):
 *             act_out bf16 [M][hidden] (dW2 = d out^T act), dpre_out bf16 [M][hidden] and xhat_out bf16 [M][C] (may be NULL)
 *             (dW1f = d pre^T xhat, un-folded by clv_ln_fold_bwd). */
int clv_mlp_fused_supported(int32_t C, int32_t hidden);
int clv_mlp_fused_fwd(const void* a, const void* res, void* sum_out, float* mean, float* rstd, const void* w1f,
                      const float* b1f, const void* w2, const float* b2, void* out, int64_t M, int32_t C, int32_t hidden,
                      float eps, const float* xscale, int32_t rows_per_sample, void* stream);
int clv_mlp_fused_bwd(const void* tsum, const float* mean, const float* rstd, const void* dout, const void* dsum,
                      const void* w1f, const float* b1f, const void* w2t, void* da, void* dres, void* act_out,
                      void* dpre_out, void* xhat_out, int64_t M, int32_t C, int32_t hidden, const float* xscale,
                      int32_t rows_per_sample, void* stream);

/* db[n] += sum_m dy[m][n] (bf16 dy, row stride ld; N, ld multiples of 8): the bias gradient of the
 * library-GEMM Linear layers (BERT / fusion / MLM head), ACCUMULATED into db. */
int clv_colsum(const void* dy, float* db, int64_t M, int32_t N, int32_t ld, void* stream);

/* ------------------------------------------------------------------ focal MLM loss
 * SoftmaxFocalLossMultiClass.forward (mmaction/models/losses/focal_loss.py:61-72) on the
 * masked rows selected by multimodal_transformer_pretrain.py:137-139, fused:
 * rows with label == -100 are skipped; loss = mean over the others of (1-pt)^gamma * ce.
 * logits bf16 or float [rows][V] (is_bf16); labels int64 [rows]; row_ce,row_lse float
 * [rows] saved for backward; loss float [1]; count float [1] (number of masked rows) —
 * both must be ZERO on entry (they double as the accumulators). */
int clv_focal_ce_fwd(const void* logits, int32_t is_bf16, const int64_t* labels, float* row_ce,
                     float* row_lse, float* loss, float* count, int64_t rows, int32_t V,
                     float gamma, void* stream);
/* dlogits (same dtype as logits) = dloss * d loss / d logits (zero for skipped rows). */
int clv_focal_ce_bwd(const void* logits, int32_t is_bf16, const int64_t* labels, const float* row_ce,
                     const float* row_lse, const float* count, const float* dloss, void* dlogits,
                     int64_t rows, int32_t V, float gamma, void* stream);
/* The same on rows of stride ld >= V elements (logits and dlogits alike): the padded score buffer of the MLM decoder (V =
 * 30522 is not a multiple of 8, ld = 30528 is).  The backward zeroes the padding columns [V, ld) of dlogits. (ABI 8) */
int clv_focal_ce_fwd_ld(const void* logits, int32_t is_bf16, const int64_t* labels, float* row_ce, float* row_lse,
                        float* loss, float* count, int64_t rows, int32_t V, int64_t ld, float gamma, void* stream);
int clv_focal_ce_bwd_ld(const void* logits, int32_t is_bf16, const int64_t* labels, const float* row_ce,
                        const float* row_lse, const float* count, const float* dloss, void* dlogits, int64_t rows,
                        int32_t V, int64_t ld, float gamma, void* stream);

/* ------------------------------------------------------------------ exclusive InfoNCE + rank
 * ExclusiveNCEwithRankingLoss.forward after the all-gather
 * (mmaction/models/losses/contrastive_loss.py:112-161): cos_norm of the four gathered
 * embeddings, three G×G similarity matmuls / temperature, the three exclusive [G,3G] row
 * log-softmaxes, the [3G,G] column log-softmax, diagonals, MarginRankingLoss(margin).
 * e0..e3 float [G][Dm] with row stride ld floats (video, text, text_mask, text_recon) — ld > Dm lets
 * the four be slots of one packed [G][k][Dm] tensor, as the all-gather delivers them.  out float [2] =
 * {nce_loss, rank_t_tm_loss}.  work: float scratch, >= clv_infonce_work_floats(G, Dm). */
int64_t clv_infonce_work_floats(int32_t G, int32_t Dm);
int clv_infonce_fwd(const float* e0, const float* e1, const float* e2, const float* e3, float* out,
                    float* work, int32_t G, int32_t Dm, int32_t ld, float temperature, float margin,
                    void* stream);
/* dout float [2] = upstream grads of {nce, rank}; d0..d3 float [G][Dm] with row stride ldd.  `work`
 * must be the buffer the matching forward filled (e0..e3 are not re-read and may be NULL). */
int clv_infonce_bwd(const float* e0, const float* e1, const float* e2, const float* e3,
                    const float* dout, const float* work, float* d0, float* d1, float* d2, float* d3,
                    int32_t G, int32_t Dm, int32_t ldd, float temperature, float margin, void* stream);
/* The pre-training step evaluates the loss twice on slots of ONE packed fp32 [G][k][Dm] tensor — video -> text on
 * (video, text, masked text, video-recon) and text -> video on (text, video, masked video, text-recon)
 * (multimodal_transformer_pretrain.py:147-169).  The pair form runs both evaluations in the same launches (3 forward,
 * 4 backward).  slots: HOST int32 [8] = the four slot indices of evaluation 0, then of evaluation 1 (distinct within an
 * evaluation, < k).  out float [4] = {nce_0, rank_0, nce_1, rank_1}; work >= 2 * clv_infonce_work_floats(G, Dm) floats.
 * Backward: dout = HOST array of four DEVICE pointers to the fp32 upstream gradients of out[0..3] (each where autograd left
 * it; NULL = 0); dpacked [G][k][Dm] is WRITTEN in full (a slot read by both evaluations receives the sum, a slot read by
 * none zeros). */
int clv_infonce_pair_fwd(const float* packed, const int32_t* slots, float* out, float* work, int32_t G, int32_t k,
                         int32_t Dm, float temperature, float margin, void* stream);
int clv_infonce_pair_bwd(const float* const* dout, const float* work, const int32_t* slots, float* dpacked, int32_t G,
                         int32_t k, int32_t Dm, float temperature, float margin, void* stream);

/* NormSoftmaxLoss (mmaction/models/losses/contrastive_loss.py:26-68), the retrieval fine-tuning loss
 * (multimodal_transformer_finetune.py:83-86): x = normalise(video) . normalise(text)^T / temperature,
 * out[0] = -mean diag(log_softmax(x, 1)) - mean diag(log_softmax(x^T, 1)).  video/text float [G][Dm];
 * eps = the norm clamp (1e-12 for F.normalize :51-52, 1e-8 for cos_sim :10-18).  When sim_mat (float
 * [G][G]) is given it is used as x and video/text/Dm/temperature/eps are ignored (:55-56).
 * work: float scratch >= clv_normsoftmax_work_floats(G, Dm). */
int64_t clv_normsoftmax_work_floats(int32_t G, int32_t Dm);
int clv_normsoftmax_fwd(const float* video, const float* text, const float* sim_mat, float* out, float* work,
                        int32_t G, int32_t Dm, float temperature, float eps, void* stream);
/* dout float [1]; `work` is the buffer the forward filled.  Without sim_mat: dvideo/dtext float [G][Dm]
 * (dsim ignored); with sim_mat: dsim float [G][G] (dvideo/dtext ignored). */
int clv_normsoftmax_bwd(const float* sim_mat, const float* dout, const float* work, float* dvideo, float* dtext,
                        float* dsim, int32_t G, int32_t Dm, float temperature, void* stream);

/* ------------------------------------------------------------------ optimizer
 * Grad-norm (clip_grad_norm_, mmcv_Fp16OptimizerHook.py:127-137) + AdamW step on flat
 * buffers (optimizer cfg pretrain_webvid_cc3m.py:129-137).
 * clv_sumsq: acc[0] += sum(g^2) over n floats (caller zeroes acc). */
int clv_sumsq(const float* g, float* acc, int64_t n, void* stream);
/* One fused AdamW step over a flat fp32 segment.  sumsq: float[1] device = total grad norm²
 * (all segments); the clip coefficient min(1, max_norm/(sqrt(sumsq)+1e-6)) is applied on
 * device, and the step is skipped when sumsq is not finite.  shadow (bf16, may be NULL)
 * receives the updated weights for the bf16 compute path.  grad_scale multiplies g first. */
int clv_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, const float* sumsq,
                   int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_c1, float bias_c2, float max_norm, float grad_scale, void* stream);

/* ------------------------------------------------------------------ Linear layers: LDS-tiled MFMA GEMM (gemm_nt.hip)
 * C[M][N] = A[M][K] . B[N][K]^T with fused epilogues — the forward and input-gradient GEMMs of nn.Linear as the
 * reference's Mlp / qkv / proj / PatchMerging.reduction and the HF BERT layers run them
 * (mmaction/models/backbones/swin_transformer_3d.py:262-268, 376, 398, 543; transformers 4.6.1 BertIntermediate /
 * BertOutput), replacing torch's addmm + separate GELU kernels:
 *   forward   y  = x W^T + b        A = x  [M][K], B = W   [N][K]          epilogue BIAS | BIAS_GELU | NONE
 *   dgrad     dx = dy W             A = dy [M][N], B = W^T [K][N]          epilogue NONE | DGELU
 * a, b, c, c2, aux: bf16, row strides lda / ldb / ldc elements (multiples of 8; c2 and aux share ldc); bias fp32 [N].
 *   CLV_GEMM_EPI_NONE       c = acc
 *   CLV_GEMM_EPI_BIAS       c = acc + bias
 *   CLV_GEMM_EPI_BIAS_GELU  c2 = acc + bias (the pre-activation, kept for backward), c = GELU_erf(acc + bias)
 *   CLV_GEMM_EPI_DGELU      c = acc * GELU_erf'(aux)      (aux = the forward's pre-activation)
 *   CLV_GEMM_EPI_BIAS_GELU_D  c2 = GELU_erf'(acc + bias) (what the backward needs of the pre-activation, computed where its
 *                             erf / exp are in hand anyway), c = GELU_erf(acc + bias)
 *   CLV_GEMM_EPI_MUL        c = acc * aux                 (aux = that c2: the GELU backward without a transcendental)
 * Needs N % 8 == 0, K % 64 == 0, 16-byte aligned pointers (clv_gemm_nt_supported); fp32 accumulation. */
#define CLV_GEMM_EPI_NONE 0
#define CLV_GEMM_EPI_BIAS 1
#define CLV_GEMM_EPI_BIAS_GELU 2
#define CLV_GEMM_EPI_DGELU 3
#define CLV_GEMM_EPI_BIAS_GELU_D 4
#define CLV_GEMM_EPI_MUL 5
int clv_gemm_nt_supported(int64_t M, int32_t N, int32_t K);
int clv_gemm_nt(const void* a, const void* b, const float* bias, const void* aux, void* c, void* c2, int64_t M,
                int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t epilogue, void* stream);
/* The same GEMM with a work buffer (ABI 8).  Few-tile long-contraction layers (K >= 1536 and <= 256 tiles of 128 x 128:
 * fc2 / the fc1 input gradient of Swin stage 3 — swin_transformer_3d.py:262-268 —, of the fusion encoder and of the text
 * tower) cut the contraction into slices that run as workgroups of their own: each slice leaves an fp32 partial sum in
 * `work` ([slices][M][N]) and a second kernel adds the slices and applies the epilogue.  clv_gemm_nt_work_bytes(M, N, K) =
 * the bytes that plan needs (0: the shape runs in one pass — then `work` may be NULL); with work == NULL or a buffer that
 * is too small the call runs in one pass, as clv_gemm_nt does.  work: 16-byte aligned, used only inside the call. */
int64_t clv_gemm_nt_work_bytes(int64_t M, int32_t N, int32_t K);
int clv_gemm_nt_ex(const void* a, const void* b, const float* bias, const void* aux, void* c, void* c2, int64_t M,
                   int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t epilogue, void* work,
                   int64_t work_bytes, void* stream);
/* Batched transposes of bf16 matrices in one launch (the engine's W^T shadows, refreshed after every optimizer step).
 * table: n_entries x {int64 src_off, dst_off (elements from src_base / dst_base); int32 rows, cols, tile_begin,
 * tiles_c} on the device; entry e covers ceil(rows/64) * tiles_c tiles starting at tile_begin (tiles_c =
 * ceil(cols/64)); total_tiles = the grid.  dst[c][r] = src[r][c]. */
#define CLV_TRANSPOSE_ENTRY_BYTES 32
int clv_transpose_batch(const void* src_base, void* dst_base, const void* table, int32_t n_entries,
                        int32_t total_tiles, void* stream);

/* The same step with the optimizer's scalars held on the DEVICE (no host sync, hipGraph-safe):
 * state = CLV_OPTIM_STATE_BYTES bytes {float coef, bc1, bc2_sqrt, norm; int32 skip, t, skipped, pad; float loss_scale,
 * scale_factor; int32 scale_window, scale_iter, last_overflow, dynamic, pad, pad}, zeroed once by the caller.  The second
 * half is the reference's LossScaler (mmaction/core/hooks/fp16_utils.py:285-389) held on the device: with loss_scale > 0
 * the caller multiplies the root gradient of its backward by state.loss_scale (a device read: it stays inside a hipGraph),
 * clv_optim_prep divides it out of the norm and the update, and with dynamic = 1 applies update_scale (:351-362): an
 * overflow (non-finite norm) halves the scale (floor 1) and records the iteration, scale_window overflow-free iterations
 * since then grow it by scale_factor (the host sets last_overflow = -1, scale_factor and scale_window once).  clv_optim_prep (one thread, after every segment's clv_sumsq): reads and re-zeroes sumsq, computes the
 * clip coefficient (grad_scale * min(1, max_norm/(norm+1e-6)); max_norm <= 0: no clipping); a finite norm advances
 * Adam's step count t and refreshes the bias corrections, a non-finite one sets skip — the reference skips
 * optimizer.step() on overflow (mmcv_Fp16OptimizerHook.py:123-141), so Adam's step count does not advance either.
 * clv_adamw_step_dev: clv_adamw_step with coefficient / bias corrections / skip read from state. */
#define CLV_OPTIM_STATE_BYTES 64
int clv_optim_prep(float* sumsq, void* state, float beta1, float beta2, float max_norm, float grad_scale,
                   void* stream);
/* clv_optim_prep with the norm slots of clv_linear_wgrad_batch_ss / clv_wgrad_fold_batch_ss: the CLV_SUMSQ_SLOTS partial sums
 * are added to sumsq[0] and re-zeroed.  clv_sumsq_ranges: sumsq over a list of ranges of ONE buffer in one launch — table
 * (device) = n_blocks x {int64 offset, int64 count} in floats from base, one entry per 256-thread block (the host cuts the
 * ranges into chunks of <= CLV_SUMSQ_CHUNK floats; offsets are multiples of 4): acc[0] += sum g^2. */
#define CLV_SUMSQ_CHUNK 16384
int clv_optim_prep_slots(float* sumsq, float* sumsq_slots, void* state, float beta1, float beta2, float max_norm,
                         float grad_scale, void* stream);
int clv_sumsq_ranges(const float* base, const void* table, int32_t n_blocks, float* acc, void* stream);
int clv_adamw_step_dev(float* p, const float* g, float* m, float* v, void* shadow, const void* state, int64_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* ------------------------------------------------------------------ parity mode (fp32 storage + fp32 arithmetic)
 * The reference's CPU path is fp32 (north_star: "match the reference mmaction CPU path ... losses within 1e-3").  The
 * training path above computes on bf16 MFMA operands; these two entry points let the SAME host graph (registered modules,
 * window geometry, token maps, bias-table indexing, masks, heads, losses) run with fp32 storage and fp32 arithmetic so the
 * step losses can be asserted at 1e-3 against the reference goldens, and the gradients at fp32 tolerances.
 *
 * clv_sgemm_nt: C[M][N] (ldc) = A[M][K] (lda) . B[N][K]^T (ldb) + bias[N] — every nn.Linear of the path
 *   (swin_transformer_3d.py:257-259,361-366,527; transformers BertSelfAttention / BertIntermediate / BertOutput as called
 *   from bert_from_hugface.py:30, cross_transformer.py:109-110; mlm_itm_head.py:33-41) on the exact-f32 MFMA 16x16x4.
 * clv_attn_f32_fwd: clv_attn_fwd's arithmetic (both modes, same ClvAttnGeom, strides in floats) on fp32 q / k / v / o;
 *   round_p != 0 rounds the un-normalised probabilities to bf16 before P.V (emulates the MFMA kernels' P operand — used
 *   by the error-isolation runs recorded in DESIGN.md). */
int clv_sgemm_nt(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N, int32_t K,
                 int64_t lda, int64_t ldb, int64_t ldc, void* stream);
/* fp32 GEMM with general strides — the [batch, D]-sized projection heads of the contrastive losses (ssl_head.py:24-35,
 * 158-166,240-245; the reference forces fp32 there, contrastive_loss.py:102), on the training path too:
 * C[i][j] (ldc) (+)= sum_k A[i*sai + k*sak] * B[j*sbj + k*sbk] (+ bias[j]);  accumulate != 0: added into C (the fp32 gradient
 * slab).  Forward x W^T, input gradient dy W and weight gradient dy^T x are this one kernel with different strides. */
int clv_sgemm_strided(const float* A, const float* B, const float* bias, float* C, int64_t M, int32_t N, int32_t K,
                      int64_t sai, int64_t sak, int64_t sbj, int64_t sbk, int64_t ldc, int32_t accumulate, void* stream);
/* The same with rowsum[i] (+)= sum_k A[i*sai + k*sak] from the same launch (ABI 13): the weight-gradient form of a head layer
 * (A = dy^T) delivers the bias gradient with it.  Contractions K <= 96 only (the rows of a batch), else CLV_ERR_UNSUPPORTED. */
int clv_sgemm_strided_rowsum(const float* A, const float* B, float* C, float* rowsum, int64_t M, int32_t N, int32_t K,
                             int64_t sai, int64_t sak, int64_t sbj, int64_t sbk, int64_t ldc, int32_t accumulate,
                             void* stream);
int clv_attn_f32_fwd(const float* q, const float* k, const float* v, float* o, const float* bias, const int32_t* rid,
                     const float* kmask, const ClvAttnGeom* geom, int32_t round_p, void* stream);
/* Backward of clv_attn_f32_fwd in fp32 (ABI 8): dq / dk / dv with the strides of q / k / v, dbias (may be NULL) ACCUMULATED
 * into the [rows, nH] relative-position table gradient (WindowAttention3D, swin_transformer_3d.py:375-400; HF
 * BertSelfAttention in mode 0), work = clv_attn_f32_bwd_work_floats(geom) floats (lse and delta per (group, head, query)).
 * Two recompute passes, one writer per output element: parity mode checks the step's gradients with it. */
int64_t clv_attn_f32_bwd_work_floats(const ClvAttnGeom* geom);
int clv_attn_f32_bwd(const float* q, const float* k, const float* v, const float* dout, const float* bias,
                     const int32_t* rid, const float* kmask, float* dq, float* dk, float* dv, float* dbias, float* work,
                     const ClvAttnGeom* geom, void* stream);

/* ------------------------------------------------------------------ fp8 forward GEMMs (BASELINE config 5)
 * "fp8 MFMA QKV / patch-proj path": the forward GEMM of a Linear on OCP e4m3 operands with one fp32 scale per row of
 * each operand (per token for the activation, per output channel for the weight), fp32 accumulation on the K = 128 matrix
 * instruction (v_mfma_f32_16x16x128_f8f6f4, twice the bf16 MFMA rate), bf16 output; gradients stay bf16 (the backward
 * takes the bf16 activation and weight).  Replaces, under CLOVER_FP8=1, the same torch.nn.Linear sites as clv_gemm_nt
 * (swin_transformer_3d.py:257-268,361-366,527; the transformers BERT layers called from bert_from_hugface.py:30 and
 * cross_transformer.py:109-110); precision policy site: mmaction/core/hooks/fp16_utils.py:215-259.
 *
 * clv_quant_fp8_rows: q[r][k] = e4m3(x[r][k] / scale[r]), scale[r] = max_k |x[r][k]| / 448; x bf16 [rows][K] (ldx),
 *   q bytes [rows][K] (ldq); K % 8 == 0, K <= 4096.
 * clv_gemm_nt_fp8: c[M][N] bf16 = (a8[M][K] . b8[N][K]^T) * sa[m] * sb[n] (+ bias; CLV_GEMM_EPI_NONE / _BIAS / _BIAS_GELU_D
 *   with c2 = GELU'(pre) as clv_gemm_nt); K % 128 == 0, lda / ldb in bytes and % 16 == 0. */
/* clv_patch_embed_fwd with the projection on the fp8 matrix instruction: w8 [C][96] e4m3 bytes + wscale [C] (the conv
 * weight through clv_quant_fp8_rows); each token's 96 patch values are scaled by 448 / their maximum and converted to
 * e4m3 inside the kernel.  Everything else (bias, LayerNorm, blend, outputs, saved tensors) as clv_patch_embed_fwd; the
 * backward is the bf16 one. */
int clv_patch_embed_fwd_fp8(const float* x, const void* w8, const float* wscale, const float* bias, const float* gamma,
                            const float* beta, const float* mask_token, const int64_t* vmask, void* out_clean,
                            void* out_masked, void* z_out, float* mean, float* rstd, int32_t B, int32_t T, int32_t H,
                            int32_t W, int32_t C, int32_t mh, int32_t mw, float eps, void* stream);
int clv_quant_fp8_rows(const void* x, void* q, float* scale, int64_t rows, int32_t K, int64_t ldx, int64_t ldq,
                       void* stream);
int clv_gemm_nt_fp8_supported(int64_t M, int32_t N, int32_t K);
int clv_gemm_nt_fp8(const void* a8, const void* b8, const float* sa, const float* sb, const float* bias, void* c, void* c2,
                    int64_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc, int32_t epilogue,
                    void* stream);

/* Data-parallel gradient exchange in bf16 (the reference all-reduces the fp16 model gradients:
 * mmaction/core/hooks/mmcv_Fp16OptimizerHook.py:119-122): clv_pack_bf16 writes the bf16 wire copy of an fp32 gradient-slab
 * slice (RCCL all-reduces THAT over xGMI: half the bytes), clv_sumsq_bf16 / clv_adamw_step_dev_bf16g are clv_sumsq /
 * clv_adamw_step_dev reading the reduced bf16 gradient — the update itself (moments, master weights) stays fp32. */
int clv_pack_bf16(const float* src, void* dst, int64_t n, void* stream);
int clv_sumsq_bf16(const void* g, float* acc, int64_t n, void* stream);
int clv_adamw_step_dev_bf16g(float* p, const void* g, float* m, float* v, void* shadow, const void* state, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CLOVER_HIP_H */
