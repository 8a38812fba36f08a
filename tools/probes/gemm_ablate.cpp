// Stand-alone timing / ablation driver for clover_amd/csrc/gemm_nt.hip (no torch): includes the kernel source, so
// -DGN_ABL_* switches compile variants.  Build (cross-compiles without a GPU):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I clover_amd/csrc [-DGN_ABL_NOMFMA ...] tools/probes/gemm_ablate.cpp -o tools/probes/bin/gemm_ablate
#include "gemm_nt.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

struct Shape { int64_t M; int N, K; const char* name; };

int main(int argc, char** argv) {
    std::vector<Shape> shapes = {
        {50176, 576, 192, "qkv s1"}, {50176, 768, 192, "fc1 s1"}, {50176, 192, 768, "fc2 s1"},
        {12544, 1152, 384, "qkv s2"}, {12544, 384, 384, "proj s2"}, {12544, 1536, 384, "fc1 s2"}, {12544, 384, 1536, "fc2 s2"},
        {3136, 2304, 768, "qkv s3"}, {3136, 3072, 768, "fc1 s3"}, {3136, 768, 3072, "fc2 s3"},
        {3648, 3072, 768, "fc1 fu"}, {3648, 768, 3072, "fc2 fu"}, {512, 3072, 768, "fc1 bert"}, {512, 768, 3072, "fc2 bert"}};
    const int epi = argc > 1 ? atoi(argv[1]) : 1;
    const int iters = 50;
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (auto& s : shapes) {
        bf16_t *a, *b, *c, *c2, *aux;
        float* bias;
        hipMalloc(&a, s.M * s.K * 2);
        hipMalloc(&b, (size_t)s.N * s.K * 2);
        hipMalloc(&c, s.M * s.N * 2);
        hipMalloc(&c2, s.M * s.N * 2);
        hipMalloc(&aux, s.M * s.N * 2);
        hipMalloc(&bias, s.N * 4);
        std::vector<uint16_t> h(s.M * s.K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 20 & 0x3ff) - ((i & 1) << 15);
        hipMemcpy(a, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        h.resize((size_t)s.N * s.K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3a00 + (uint16_t)((i * 40503u) >> 7 & 0x1ff) - ((i & 2) << 14);
        hipMemcpy(b, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipMemset(aux, 0x3c, s.M * s.N * 2);
        hipMemset(bias, 0, s.N * 4);
        for (int w = 0; w < 3; ++w) clv_gemm_nt(a, b, bias, aux, c, c2, s.M, s.N, s.K, s.K, s.K, s.N, epi, st);
        hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        int rc = 0;
        for (int i = 0; i < iters; ++i) rc |= clv_gemm_nt(a, b, bias, aux, c, c2, s.M, s.N, s.K, s.K, s.K, s.N, epi, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 2.0 * s.M * s.N * s.K;
        printf("%-9s M=%6ld N=%5d K=%5d: %7.1f us  %6.0f TF  rc=%d\n", s.name, (long)s.M, s.N, s.K, us, fl / us / 1e6, rc);
        hipFree(a); hipFree(b); hipFree(c); hipFree(c2); hipFree(aux); hipFree(bias);
    }
    return 0;
}
