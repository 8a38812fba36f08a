// Issue rate of v_mfma_f32_16x16x32_bf16 vs v_mfma_f32_32x32x16_bf16 on one SIMD: W waves per SIMD, each a loop of N independent
// MFMAs per trip; prints s_memtime cycles per MFMA and the implied fraction of the dense bf16 peak (1024 flop / clk / SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.cpp -o tools/probes/bin/mfma_rate && tools/probes/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

template <int NACC>
__global__ void __launch_bounds__(512, 2) k16(float* out, unsigned long long* cyc, int trips) {
    f32x4_t acc[NACC];
    bf16x8_t a[6], b[6];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 6; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(float)(threadIdx.x + i + e); b[i][e] = (__bf16)(float)(threadIdx.x * 3 + i - e); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i % 6], b[(i / 6) % 6], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void __launch_bounds__(512, 2) k32(float* out, unsigned long long* cyc, int trips) {
    f32x16_t acc[NACC];
    bf16x8_t a[3], b[3];
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int i = 0; i < 3; ++i)
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(float)(threadIdx.x + i + e); b[i][e] = (__bf16)(float)(threadIdx.x * 3 + i - e); }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % 3], b[(i / 3) % 3], acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 1024 * 512 * 4);
    hipMalloc(&cyc, 1024 * 8);
    const int trips = 2000;
    for (int threads : {256, 512}) {
        for (int blocks : {1, 256}) {
            for (int kind = 0; kind < 2; ++kind) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (kind == 0) hipLaunchKernelGGL(k16<36>, dim3(blocks), dim3(threads), 0, 0, out, cyc, trips);
                    else hipLaunchKernelGGL(k32<9>, dim3(blocks), dim3(threads), 0, 0, out, cyc, trips);
                    hipEventRecord(e1);
                    hipDeviceSynchronize();
                }
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long c;
                hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const int n = kind == 0 ? 36 : 9;
                const double per = (double)c / trips / n, flop = kind == 0 ? 16384.0 : 32768.0;
                const int wps = threads / 256;
                printf("%s  %d waves/SIMD  %3d blocks: %.1f cycles per MFMA per wave -> %.0f flop/clk/SIMD (%.0f %% of 1024); wall %.3f ms -> %.0f TFLOP/s chip-equivalent at 256 CUs\n",
                       kind == 0 ? "16x16x32" : "32x32x16", wps, blocks, per, flop * wps / per, 100.0 * flop * wps / per / 1024.0, ms,
                       (double)blocks * (threads / 64) * trips * n * flop / (ms * 1e-3) / 1e12 * (256.0 / blocks));
            }
        }
    }
    return 0;
}
