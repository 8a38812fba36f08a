#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 40];   // 64 rows x 40 cols (row stride 80 B)
    for (int i = threadIdx.x; i < 64 * 40; i += 64) lds[i] = (uint16_t)((i / 40) * 100 + (i % 40));   // value = row*100 + col
    __syncthreads();
    int l = threadIdx.x, lg = l >> 4, lr = l & 15;
    // block: rows lg*4 .. +4, cols 16..32 ; lane lr points at (row lg*4 + lr/4, col 16 + (lr%4)*4)
    v4s* p = (v4s*)&lds[(lg * 4 + (lr >> 2)) * 40 + 16 + (lr & 3) * 4];
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        int expect = ((l >> 4) * 4 + j) * 100 + 16 + (l & 15);
        if (h[l * 4 + j] != expect) { if (bad < 8) printf("lane %d j %d got %d expect %d\n", l, j, h[l*4+j], expect); ++bad; }
    }
    printf("bad=%d\n", bad);
    for (int l = 0; l < 20; ++l) printf("lane %d: %d %d %d %d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
