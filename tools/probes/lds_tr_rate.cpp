// Throughput of ds_read_b64_tr_b16 (the transposing fragment read of the weight-gradient kernels): W waves of one workgroup,
// each a loop of NR reads from a swizzled [32][W16 x 16] bf16 image followed by s_waitcnt lgkmcnt(0); cycles per trip and
// bytes per clock per CU.  Compared with ds_read_b64 and ds_read_b128 at the same addresses.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_tr_rate.cpp -o tools/probes/bin/lds_tr_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s_t __attribute__((ext_vector_type(4)));
template <int KIND, int NR>
__global__ void __launch_bounds__(512) k(unsigned* out, unsigned long long* cyc, int trips, int w) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int lane = threadIdx.x & 63, lg = lane >> 4, lr = lane & 15, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 36 * 1024 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(smem)[i] = i * 2654435761u;
    __syncthreads();
    const int t = __builtin_ctz(w | 8);
    const int r = lg * 4 + (lr >> 2), s = (r >> (3 - t)) & ((1 << t) - 1);
    unsigned addr[NR];
    for (int i = 0; i < NR; ++i) {
        const int u = ((wave * 3 + i / 2) % w) ^ s;
        addr[i] = (unsigned)(((r + (i & 1) * 16) * w + u) * 32 + (lr & 3) * 8);
        if (KIND == 2) addr[i] = (unsigned)(((lane >> 1) * w + u) * 32 + (lane & 1) * 16);
    }
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int tr = 0; tr < trips; ++tr) {
        uint2 v[NR];
        uint4 q[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (KIND == 0) {
                const v4s_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(smem + addr[i]));
                union { v4s_t v; uint2 u; } cv; cv.v = x; v[i] = cv.u;
            } else if (KIND == 1) {
                const volatile uint2* p = reinterpret_cast<const volatile uint2*>(smem + addr[i]);
                v[i] = make_uint2(p->x, p->y);
            } else {
                const volatile uint4* p = reinterpret_cast<const volatile uint4*>(smem + addr[i]);
                q[i] = make_uint4(p->x, p->y, p->z, p->w);
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) acc += KIND == 2 ? (q[i].x ^ q[i].w) : (v[i].x ^ v[i].y);
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND>
void run(const char* name, int waves, int w, unsigned* out, unsigned long long* cyc) {
    const int trips = 2000;
    constexpr int NR = 24;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<KIND, NR>), dim3(1), dim3(64 * waves), 36 * 1024, 0, out, cyc, trips, w);
        (void)hipDeviceSynchronize();
    }
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)c / trips, bytes = (KIND == 2 ? 1024.0 : 512.0) * NR * waves;
    printf("%-22s %d waves, row = %2d groups: %.0f cycles per %d reads per wave = %.1f cycles per wave-instruction, %.0f B/clk/CU\n", name, waves,
           w, per, NR, per / NR, bytes / per);
}
int main() {
    unsigned* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 8);
    for (int w : {12, 24, 8})
        for (int waves : {1, 4, 8}) {
            run<0>("ds_read_b64_tr_b16", waves, w, out, cyc);
            run<1>("ds_read_b64", waves, w, out, cyc);
            run<2>("ds_read_b128", waves, w, out, cyc);
        }
    return 0;
}
