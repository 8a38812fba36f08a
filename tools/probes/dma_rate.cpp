// LDS-DMA (global_load_lds_dwordx4) throughput per access pattern, data L2 / MALL resident.
// Each wave issues PIECES 1-KiB pieces per iteration into its own LDS area, waits vmcnt(0) every UNROLL pieces.
// pattern: bytes contiguous per row (64, 128, 256, 1024) -> rows per piece = 1024 / rowbytes; row stride = ld bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

__device__ __forceinline__ void dma16(const void* src, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64) dma_kernel(const char* base, int rowbytes, int64_t ld, int64_t span, int iters,
                                                         int inflight, int* sink) {
    __shared__ __attribute__((aligned(1024))) char lds[WAVES * 8 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&lds[0]) + wave * 8192;
    const int lanes_per_row = rowbytes / 16;
    const int r = lane / lanes_per_row, c = lane % lanes_per_row;
    const int rows_per_piece = 64 / lanes_per_row;
    // each workgroup walks its own window of the buffer (span bytes, reused -> L2 resident)
    int64_t off = ((int64_t)blockIdx.x * 7919 * 4096) % span;
    const char* p0 = base + (int64_t)r * ld + c * 16 + wave * (int64_t)rows_per_piece * ld;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char* p = p0 + off;
            dma16(p, __builtin_amdgcn_readfirstlane(lbase + u * 1024));
            off += (int64_t)WAVES * rows_per_piece * ld;
            if (off >= span) off -= span;
        }
        if (inflight == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && lds[17] == 123) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int64_t bufbytes = 1ll << 30;
    char* buf;
    int* sink;
    hipMalloc(&buf, bufbytes);
    hipMalloc(&sink, 4);
    hipMemset(buf, 1, bufbytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 200;
    struct Cfg { int rowbytes; int64_t ld; int64_t span; const char* what; };
    Cfg cfgs[] = {{64, 768, 8ll << 20, "64B rows ld768, 8MB (L2)"},   {128, 768, 8ll << 20, "128B rows ld768, 8MB (L2)"},
                  {256, 768, 8ll << 20, "256B rows ld768, 8MB (L2)"}, {1024, 1024, 8ll << 20, "linear 1KB, 8MB (L2)"},
                  {64, 768, 96ll << 20, "64B rows, 96MB (MALL)"},     {128, 768, 96ll << 20, "128B rows, 96MB (MALL)"},
                  {1024, 1024, 96ll << 20, "linear, 96MB (MALL)"},    {128, 768, 1000ll << 20, "128B rows, 1GB (HBM)"},
                  {1024, 1024, 1000ll << 20, "linear, 1GB (HBM)"}};
    for (int wgs_per_cu : {1, 2}) for (int inflight : {0, 1}) for (auto& c : cfgs) {
        const int grid = 256 * wgs_per_cu;
        hipLaunchKernelGGL(dma_kernel<4>, dim3(grid), dim3(256), 0, 0, buf, c.rowbytes, c.ld, c.span, 20, inflight, sink);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(dma_kernel<4>, dim3(grid), dim3(256), 0, 0, buf, c.rowbytes, c.ld, c.span, iters, inflight, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * 4 * iters * 8 * 1024;
        printf("%d WG/CU x4 waves, %s-deep: %-28s %7.2f TB/s  %6.1f B/clk/CU(2.4GHz)  %6.0f cyc/piece/wave\n", wgs_per_cu,
               inflight ? "16" : "8", c.what, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9,
               ms * 1e-3 * 2.4e9 / (iters * 8));
    }
    return 0;
}
