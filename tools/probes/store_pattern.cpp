// Store-throughput probe: a [M][N] bf16 matrix written tile by tile (128 x 128 per 256-thread workgroup, 2 x 2 waves of
// 64 x 64) with the wave-instruction footprints a GEMM epilogue can produce:
//   mode 0: 16 rows x 64 B  (gemm_nt today: 4 lanes x 16 B per row)      mode 1: 8 rows x 128 B (full lines)
//   mode 2: 4 rows x 256 B                                                mode 3: 1 KiB contiguous (fill-like reference)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void __launch_bounds__(256) k(uint16_t* c, int64_t M, int N, int mode, int tilesN, int ntiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t m0 = (int64_t)(t / tilesN) * 128 + (wave >> 1) * 64;
        const int n0 = (t % tilesN) * 128 + (wave & 1) * 64;
        const uint4 v = make_uint4(t, lane, wave, 7);
        if (mode == 3) {
            uint16_t* p = c + ((int64_t)t * 4 + wave) * 4096;      // this wave's 8 KiB, contiguous
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(p + i * 512 + lane * 8) = v;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {                               // 8 instructions x 1 KiB = the wave's 64 x 64 block
            int r, cb;                                              // row in block, 16-byte chunk in the row (0..7)
            if (mode == 0) { r = (i >> 1) * 16 + (lane & 15); cb = (i & 1) * 4 + (lane >> 4); }
            else if (mode == 1) { r = i * 8 + (lane >> 3); cb = lane & 7; }
            else { r = i * 8 + (lane >> 3); cb = lane & 7; }
            *reinterpret_cast<uint4*>(c + (m0 + r) * N + n0 + cb * 8) = v;
        }
    }
}
int main() {
    const int64_t M = 12544; const int N = 1536;
    uint16_t* c; hipMalloc(&c, M * N * 2 * 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tilesN = N / 128, ntiles = (int)(M / 128) * tilesN;
    for (int grid : {512, 1024, ntiles})
        for (int mode : {0, 1, 3}) {
            for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, c, M, N, mode, tilesN, ntiles);
            hipEventRecord(e0);
            for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, c, M, N, mode, tilesN, ntiles);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("grid %5d mode %d: %6.1f us  %5.2f TB/s\n", grid, mode, ms * 50, M * N * 2.0 / (ms / 20 * 1e-3) / 1e12);
        }
    return 0;
}
