// Optimizer step on flat fp32 buffers: global grad-norm (sum of squares) and a fused
// clip + AdamW update that also refreshes the bf16 compute copy of the weights.
// Pure HBM streaming: 16 B per lane, grid-stride.
#include "common.hpp"
#include "../../include/clover_hip.h"

namespace {

// Gradient element type of the optimizer kernels: fp32 (the slabs the backward accumulates into) or bf16 (the wire copy
// a data-parallel job all-reduces over xGMI — half the bytes; the update itself stays fp32: clv_pack_bf16,
// clv_sumsq_bf16, clv_adamw_step_dev_bf16g).
__device__ __forceinline__ float4 load_g4(const float* g, int64_t i) { return reinterpret_cast<const float4*>(g)[i]; }
__device__ __forceinline__ float4 load_g4(const bf16_t* g, int64_t i) {
    const uint2 u = reinterpret_cast<const uint2*>(g)[i];
    return make_float4(half_lo(u.x), half_hi(u.x), half_lo(u.y), half_hi(u.y));
}
__device__ __forceinline__ float load_g1(const float* g, int64_t i) { return g[i]; }
__device__ __forceinline__ float load_g1(const bf16_t* g, int64_t i) { return bf2f(g[i]); }

template <typename GT>
__global__ void __launch_bounds__(256) sumsq_kernel(const GT* __restrict__ g, float* __restrict__ acc, int64_t n) {
    __shared__ float sh[4];
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = load_g4(g, i);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
        const float v = load_g1(g, n4 * 4 + threadIdx.x);
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}

// Sum of squares over a LIST of ranges of one fp32 buffer (round 6: what is left of the gradient slabs once the kernels that
// write the first-touch weight gradients have contributed their own sums — clv_linear_wgrad_batch_ss): the table holds one
// {int64 offset, int64 count} pair per BLOCK (the host cuts every range into chunks of <= CLV_SUMSQ_CHUNK floats; offsets
// and counts are multiples of 4 except a range's tail), 16-byte loads, one atomic per block.
__global__ void __launch_bounds__(256) sumsq_ranges_kernel(const float* __restrict__ base, const int64_t* __restrict__ table,
                                                           float* __restrict__ acc) {
    __shared__ float sh[4];
    const int64_t off = table[2 * blockIdx.x], n = table[2 * blockIdx.x + 1];
    const float* g = base + off;
    const int64_t n4 = n / 4;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    if (threadIdx.x < (n - n4 * 4)) {
        const float v = g[n4 * 4 + threadIdx.x];
        s += v * v;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        if (t != 0.f) atomicAdd(acc, t);
    }
}

struct AdamArgs {
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, max_norm, grad_scale;
};

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
    g *= coef;
    p *= (1.f - a.lr * a.wd);
    m = a.beta1 * m + (1.f - a.beta1) * g;
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
    return p;
}

__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ shadow, const float* __restrict__ sumsq,
                                                    int64_t n, AdamArgs a) {
    float coef = a.grad_scale;
    if (sumsq) {
        const float ss = sumsq[0] * a.grad_scale * a.grad_scale;
        if (!(ss == ss) || ss > 3.0e38f) return;               // non-finite grad norm: skip the step
        if (a.max_norm > 0.f) {
            const float c = a.max_norm / (sqrtf(ss) + 1e-6f);  // clip_grad_norm_
            coef *= fminf(c, 1.0f);
        }
    }
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pv.x, gv.x, mv.x, vv.x, a, coef);
        adam_one(pv.y, gv.y, mv.y, vv.y, a, coef);
        adam_one(pv.z, gv.z, mv.z, vv.z, a, coef);
        adam_one(pv.w, gv.w, mv.w, vv.w, a, coef);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (shadow) {
            uint2 o;
            o.x = pack2bf(pv.x, pv.y);
            o.y = pack2bf(pv.z, pv.w);
            reinterpret_cast<uint2*>(shadow)[i] = o;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        float pv = p[i], mv = m[i], vv = v[i];
        adam_one(pv, g[i], mv, vv, a, coef);
        p[i] = pv; m[i] = mv; v[i] = vv;
        if (shadow) shadow[i] = f2bf(pv);
    }
}

// Device-resident optimizer state (clv_optim_prep writes it, clv_adamw_step_dev reads it): the clip coefficient, Adam's
// bias corrections and Adam's OWN step count t, which advances only on steps that are taken — the reference skips
// optimizer.step() on a non-finite gradient (mmcv_Fp16OptimizerHook.py:123-141), so its Adam state['step'] does not
// move there either — all without a host round trip.
struct OptimState {
    float coef;        // grad_scale * min(1, max_norm / (norm + 1e-6))
    float bc1;         // 1 - beta1^t
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    float norm;        // global gradient norm of this step (after grad_scale), for logging
    int skip;          // 1: non-finite norm, leave parameters and moments alone
    int t;             // Adam steps taken so far
    int skipped;       // steps skipped so far
    int pad;
    // The loss scaler (fp16_utils.py:285-389 LossScaler), also on the device: the backward's root gradient is multiplied by
    // loss_scale (read from here by the host graph), clv_optim_prep divides it out again and — in dynamic mode — moves it.
    float loss_scale;      // current scale; 0: no scaler in use (the caller folds any static scale into grad_scale)
    float scale_factor;    // dynamic: the factor the scale moves by (reference default 2)
    int scale_window;      // dynamic: overflow-free iterations before the scale grows (reference default 1000)
    int scale_iter;        // LossScaler.cur_iter
    int last_overflow;     // LossScaler.last_overflow_iter (host initialises it to -1)
    int dynamic;           // 1: LossScaler(mode='dynamic'); 0: static
    int pad2[2];
};

__global__ void optim_prep_kernel(float* __restrict__ acc, OptimState* __restrict__ st, float beta1, float beta2,
                                  float max_norm, float grad_scale, float* __restrict__ slots) {
    // slots (clv_optim_prep_slots): the CLV_SUMSQ_SLOTS partial sums the weight-gradient kernels left (ssq_commit), one per
    // lane of this single wave; read, re-zeroed and added to acc[0]
    float extra = 0.f;
    if (slots) {
        extra = slots[threadIdx.x * 16];
        slots[threadIdx.x * 16] = 0.f;
        extra = wave_sum(extra);
    }
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    OptimState o = *st;
    if (o.loss_scale > 0.f) grad_scale /= o.loss_scale;        // (a power of two unless the caller chose otherwise: exact)
    const float raw = acc[0] + extra;
    const float ss = raw * grad_scale * grad_scale;
    acc[0] = 0.f;                                              // ready for the next step's clv_sumsq calls
    const bool overflow = !(ss == ss) || ss > 3.0e38f || raw > 3.0e38f;
    if (overflow) {
        o.skip = 1;
        o.skipped += 1;
        o.norm = ss;
    } else {
        o.skip = 0;
        o.t += 1;
        o.norm = sqrtf(ss);
        float c = grad_scale;
        if (max_norm > 0.f) c *= fminf(max_norm / (o.norm + 1e-6f), 1.0f);      // clip_grad_norm_
        o.coef = c;
        o.bc1 = (float)(1.0 - pow((double)beta1, (double)o.t));
        o.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)o.t));
    }
    if (o.dynamic && o.loss_scale > 0.f) {                     // LossScaler.update_scale (fp16_utils.py:351-362)
        if (overflow) {
            o.loss_scale = fmaxf(o.loss_scale / o.scale_factor, 1.f);
            o.last_overflow = o.scale_iter;
        } else if (o.scale_window > 0 && (o.scale_iter - o.last_overflow) % o.scale_window == 0) {
            o.loss_scale *= o.scale_factor;
        }
        o.scale_iter += 1;
    }
    *st = o;
}

template <typename GT>
__global__ void __launch_bounds__(256) adamw_dev_kernel(float* __restrict__ p, const GT* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        bf16_t* __restrict__ shadow, const OptimState* __restrict__ st,
                                                        int64_t n, AdamArgs a) {
    if (st->skip) return;
    const float coef = st->coef;
    a.bc1 = st->bc1;
    a.bc2_sqrt = st->bc2_sqrt;
    const int64_t n4 = n / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = load_g4(g, i);
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        adam_one(pv.x, gv.x, mv.x, vv.x, a, coef);
        adam_one(pv.y, gv.y, mv.y, vv.y, a, coef);
        adam_one(pv.z, gv.z, mv.z, vv.z, a, coef);
        adam_one(pv.w, gv.w, mv.w, vv.w, a, coef);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (shadow) {
            uint2 o;
            o.x = pack2bf(pv.x, pv.y);
            o.y = pack2bf(pv.z, pv.w);
            reinterpret_cast<uint2*>(shadow)[i] = o;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
        const int64_t i = n4 * 4 + threadIdx.x;
        float pv = p[i], mv = m[i], vv = v[i];
        adam_one(pv, load_g1(g, i), mv, vv, a, coef);
        p[i] = pv; m[i] = mv; v[i] = vv;
        if (shadow) shadow[i] = f2bf(pv);
    }
}

// fp32 gradient slab slice -> its bf16 wire copy (one 16-byte load, one 8-byte store per thread)
__global__ void __launch_bounds__(256) pack_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t n4 = n / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        reinterpret_cast<uint2*>(dst)[i] = make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) dst[n4 * 4 + threadIdx.x] = f2bf(src[n4 * 4 + threadIdx.x]);
}

// Streaming kernels: ONE float4 per thread for the optimizer update (160 M parameters: 785 us = 6.1 TB/s of its 30 B per
// parameter, against 930-1050 us for 2048 persistent blocks striding 8 MB apart — tools/probes/adam_rate.py); the
// norm kernel ends in one atomic per block and keeps a bounded grid.
int grid_for(int64_t n4, int64_t cap = (1 << 20)) {
    int64_t b = (n4 + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int clv_abi_version(void) { return CLV_ABI_VERSION; }
extern "C" int clv_half_type(void) { return CLV_HALF_IS_F16; }

extern "C" int clv_sumsq(const float* g, float* acc, int64_t n, void* stream) {
    if (!g || !acc || n < 0) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if (((uintptr_t)g) & 15) return CLV_ERR_ARG;
    static const int sumsq_cap = getenv("CLV_SUMSQ_GRID") ? atoi(getenv("CLV_SUMSQ_GRID")) : 2048;
    hipLaunchKernelGGL(sumsq_kernel<float>, dim3(grid_for(n / 4, sumsq_cap)), dim3(256), 0, (hipStream_t)stream, g, acc, n);
    return clv_check_launch();
}

extern "C" int clv_sumsq_bf16(const void* g, float* acc, int64_t n, void* stream) {
    if (!g || !acc || n < 0) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if (((uintptr_t)g) & 7) return CLV_ERR_ARG;
    hipLaunchKernelGGL(sumsq_kernel<bf16_t>, dim3(grid_for(n / 4, 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)g, acc, n);
    return clv_check_launch();
}

extern "C" int clv_pack_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst || n < 0) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if ((((uintptr_t)src) & 15) || (((uintptr_t)dst) & 7)) return CLV_ERR_ARG;
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    return clv_check_launch();
}

extern "C" int clv_adamw_step(float* p, const float* g, float* m, float* v, void* shadow, const float* sumsq,
                              int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                              float bias_c1, float bias_c2, float max_norm, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || n < 0 || bias_c1 <= 0.f || bias_c2 <= 0.f) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return CLV_ERR_ARG;
    if (shadow && (((uintptr_t)shadow) & 7)) return CLV_ERR_ARG;
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, bias_c1, sqrtf(bias_c2), max_norm, grad_scale};
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (bf16_t*)shadow, sumsq, n, a);
    return clv_check_launch();
}

extern "C" int clv_optim_prep(float* sumsq, void* state, float beta1, float beta2, float max_norm, float grad_scale,
                              void* stream) {
    return clv_optim_prep_slots(sumsq, nullptr, state, beta1, beta2, max_norm, grad_scale, stream);
}

extern "C" int clv_optim_prep_slots(float* sumsq, float* sumsq_slots, void* state, float beta1, float beta2,
                                    float max_norm, float grad_scale, void* stream) {
    if (!sumsq || !state || (((uintptr_t)state) & 3)) return CLV_ERR_ARG;
    static_assert(sizeof(OptimState) == CLV_OPTIM_STATE_BYTES, "OptimState layout is part of the ABI");
    static_assert(CLV_SUMSQ_SLOTS == 64, "one slot per lane of the prep kernel's wave");
    hipLaunchKernelGGL(optim_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sumsq, (OptimState*)state, beta1,
                       beta2, max_norm, grad_scale, sumsq_slots);
    return clv_check_launch();
}

extern "C" int clv_sumsq_ranges(const float* base, const void* table, int32_t n_blocks, float* acc, void* stream) {
    if (!base || !table || !acc || n_blocks < 0 || (((uintptr_t)base) & 15) || (((uintptr_t)table) & 7)) return CLV_ERR_ARG;
    if (n_blocks == 0) return CLV_OK;
    hipLaunchKernelGGL(sumsq_ranges_kernel, dim3((unsigned)n_blocks), dim3(256), 0, (hipStream_t)stream, base,
                       (const int64_t*)table, acc);
    return clv_check_launch();
}

extern "C" int clv_adamw_step_dev(float* p, const float* g, float* m, float* v, void* shadow, const void* state,
                                  int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                  void* stream) {
    if (!p || !g || !m || !v || !state || n < 0) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return CLV_ERR_ARG;
    if (shadow && (((uintptr_t)shadow) & 7)) return CLV_ERR_ARG;
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, 1.f, 1.f, 0.f, 1.f};
    hipLaunchKernelGGL(adamw_dev_kernel<float>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (bf16_t*)shadow, (const OptimState*)state, n, a);
    return clv_check_launch();
}

extern "C" int clv_adamw_step_dev_bf16g(float* p, const void* g, float* m, float* v, void* shadow, const void* state,
                                        int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        void* stream) {
    if (!p || !g || !m || !v || !state || n < 0) return CLV_ERR_ARG;
    if (n == 0) return CLV_OK;
    if ((((uintptr_t)p) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return CLV_ERR_ARG;
    if ((((uintptr_t)g) & 7) || (shadow && (((uintptr_t)shadow) & 7))) return CLV_ERR_ARG;
    AdamArgs a{lr, beta1, beta2, eps, weight_decay, 1.f, 1.f, 0.f, 1.f};
    hipLaunchKernelGGL(adamw_dev_kernel<bf16_t>, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, p,
                       (const bf16_t*)g, m, v, (bf16_t*)shadow, (const OptimState*)state, n, a);
    return clv_check_launch();
}
