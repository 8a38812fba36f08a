// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source (gfx950)?
// If it does, the pieces of one stage need ONE M0 value + immediates instead of an M0 rewrite per piece — and the M0
// rewrite is what serialises them (each s_mov m0 waits for the previous LDS-DMA instruction to have read M0).
// Variant A: M0 rewritten per piece.  Variant B: one M0, pieces at offset:0, 1024, 2048, 3072 with the lane's global offset
// lowered by the same amount.  Prints correctness of the LDS image and the issue cycles per piece.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/dma_offset.cpp -o tools/probes/bin/dma_offset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

template <int VAR>
__global__ void __launch_bounds__(256) probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, unsigned long long* cyc, int reps) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[8 * 4096];      // 8 slots of 4 pieces
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)&lds[0] + wave * 4096;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(base);
    // piece j of this wave = the 1 KiB at src + (blockIdx * 4 + wave) * 4096 + j * 1024
    const uint32_t* gbase = src + (size_t)blockIdx.x * 4 * 1024 + 1024;      // + 4 KiB: keeps every lane offset non-negative
    unsigned v[4];
    for (int j = 0; j < 4; ++j) v[j] = (unsigned)(wave * 4096 + j * 1024 + lane * 16 - (VAR == 1 ? j * 1024 : 0));
    unsigned long long t0 = 0, t1 = 0;
    for (int r = 0; r < reps; ++r) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        unsigned keep;
        if (VAR == 0) {
            asm volatile(
                "s_mov_b32 %[keep], m0\n\t"
                "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v0], %[b]\n\t"
                "s_add_u32 m0, %[lds], 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v1], %[b]\n\t"
                "s_add_u32 m0, %[lds], 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v2], %[b]\n\t"
                "s_add_u32 m0, %[lds], 0xc00\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[v3], %[b]\n\t"
                "s_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep)
                : [lds] "s"(sbase), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [b] "s"(gbase)
                : "memory", "scc");
        } else {
            asm volatile(
                "s_mov_b32 %[keep], m0\n\t"
                "s_mov_b32 m0, %[lds]\n\ts_nop 0\n\t"
                "global_load_lds_dwordx4 %[v0], %[b]\n\t"
                "global_load_lds_dwordx4 %[v1], %[b] offset:1024\n\t"
                "global_load_lds_dwordx4 %[v2], %[b] offset:2048\n\t"
                "global_load_lds_dwordx4 %[v3], %[b] offset:3072\n\t"
                "s_mov_b32 m0, %[keep]"
                : [keep] "=&s"(keep)
                : [lds] "s"(sbase), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [b] "s"(gbase)
                : "memory", "scc");
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // dump this wave's 4 KiB
    const uint32_t* l = reinterpret_cast<const uint32_t*>(lds + wave * 4096);
    for (int i = lane; i < 1024; i += 64) out[((size_t)blockIdx.x * 4 + wave) * 1024 + i] = l[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    const int blocks = 512;
    const size_t n = (size_t)blocks * 4 * 1024;      // dwords; the source carries 4 KiB of slack in front (see gbase)
    std::vector<uint32_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u + 12345u);
    uint32_t *src, *out;
    unsigned long long* cyc;
    uint32_t* src0;
    hipMalloc(&src0, n * 4 + 4096); hipMalloc(&out, n * 4); hipMalloc(&cyc, 8);
    src = src0;
    hipMemcpy(src0 + 1024, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int var = 0; var < 2; ++var) {
        hipMemset(out, 0, n * 4);
        if (var == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, 20);
        else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, 20);
        hipDeviceSynchronize();
        std::vector<uint32_t> o(n);
        unsigned long long c = 0;
        hipMemcpy(o.data(), out, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += o[i] != h[i];
        printf("variant %c (%s): mismatching dwords %zu of %zu; issue of 4 pieces = %llu cycles (%llu per piece)\n", 'A' + var,
               var ? "one M0 + instruction offsets" : "M0 rewritten per piece", bad, n, c, c / 4);
    }
    return 0;
}
