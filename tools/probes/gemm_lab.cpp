// Stand-alone lab for clover_amd/csrc/gemm_nt.hip (no torch): includes the kernel source with -DGN_TRACE, so every launch
// leaves per-workgroup cycle sums of the main loop's phases (wait for the DMA, barrier, DMA issue, LDS reads + MFMAs,
// epilogue).  Weights rotate over copies that together exceed the Infinity Cache ("cold W", as in the step); the
// activation stays hot.  Build (cross-compiles without a GPU):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DGN_TRACE -I clover_amd/csrc -I include \
//         tools/probes/gemm_lab.cpp -o tools/probes/bin/gemm_lab
// Run:  CLV_GEMM_TILE=... CLV_GEMM_ROT=... tools/probes/bin/gemm_lab [epilogue]
#include "gemm_nt.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>

struct Shape { int64_t M; int N, K; const char* name; };

int main(int argc, char** argv) {
    std::vector<Shape> shapes = {
        {50176, 192, 768, "fc2 s1"}, {12544, 1536, 384, "fc1 s2"}, {12544, 384, 1536, "fc2 s2"},
        {3136, 2304, 768, "qkv s3"}, {3136, 3072, 768, "fc1 s3"}, {3136, 768, 3072, "fc2 s3"}, {3136, 768, 2304, "dqkv s3"},
        {3136, 768, 1536, "merge s3"}, {3648, 768, 3072, "fc2 fu"}, {512, 3072, 768, "fc1 bert"}, {512, 768, 768, "out bert"},
        {512, 768, 3072, "fc2 bert"}, {512, 768, 2304, "dqkv bert"}};
    const int epi = argc > 1 ? atoi(argv[1]) : 1;
    hipStream_t st;
    hipStreamCreate(&st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    unsigned long long* trace;
    const int max_wg = 4096;
    hipMalloc(&trace, max_wg * 16 * sizeof(unsigned long long));
    hipMemcpyToSymbol(HIP_SYMBOL(gn_trace_buf), &trace, sizeof(trace));
    for (auto& s : shapes) {
        const size_t wbytes = (size_t)s.N * s.K * 2;
        const int ncopy = (int)std::max<size_t>(8, std::min<size_t>(160, (700u << 20) / wbytes + 1));
        bf16_t *a, *b, *c, *c2, *aux;
        float* bias;
        hipMalloc(&a, s.M * s.K * 2);
        hipMalloc(&b, wbytes * ncopy);
        hipMalloc(&c, s.M * s.N * 2);
        hipMalloc(&c2, s.M * s.N * 2);
        hipMalloc(&aux, s.M * s.N * 2);
        hipMalloc(&bias, s.N * 4);
        std::vector<uint16_t> h(s.M * s.K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 20 & 0x3ff) - ((i & 1) << 15);
        hipMemcpy(a, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        h.resize((size_t)s.N * s.K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3a00 + (uint16_t)((i * 40503u) >> 7 & 0x1ff) - ((i & 2) << 14);
        for (int k = 0; k < ncopy; ++k) hipMemcpy((char*)b + k * wbytes, h.data(), wbytes, hipMemcpyHostToDevice);
        hipMemset(aux, 0x3c, s.M * s.N * 2);
        hipMemset(bias, 0, s.N * 4);
        const int64_t work_bytes = clv_gemm_nt_work_bytes(s.M, s.N, s.K);
        void* work = nullptr;
        if (work_bytes > 0) hipMalloc(&work, work_bytes);
        auto run = [&](int i) {
            return clv_gemm_nt_ex(a, (char*)b + (size_t)(i % ncopy) * wbytes, bias, aux, c, c2, s.M, s.N, s.K, s.K, s.K, s.N, epi, work,
                                  work_bytes, st);
        };
        for (int w = 0; w < 3; ++w) run(w);
        hipStreamSynchronize(st);
        const int iters = ncopy;
        hipEventRecord(e0, st);
        int rc = 0;
        for (int i = 0; i < iters; ++i) rc |= run(i + 3);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / iters, fl = 2.0 * s.M * s.N * s.K;
        // phase sums of the LAST launch (cold W), averaged over the workgroups that ran stages
        hipMemset(trace, 0, max_wg * 16 * sizeof(unsigned long long));
        run(1);
        hipStreamSynchronize(st);
        std::vector<unsigned long long> t(max_wg * 16);
        hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost);
        double sum[16] = {0};
        int n = 0;
        unsigned long long first = ~0ull, last = 0, maxtot = 0;
        for (int w = 0; w < max_wg; ++w) {
            if (!t[w * 16 + 5]) continue;
            ++n;
            for (int k = 0; k < 16; ++k) if (k != 7) sum[k] += (double)t[w * 16 + k];
            first = std::min(first, t[w * 16 + 7]);
            last = std::max(last, t[w * 16 + 7] + t[w * 16 + 6]);
            maxtot = std::max(maxtot, t[w * 16 + 6]);
        }
        const double stg = sum[5] / std::max(n, 1);
        printf("%-9s %6ldx%5dx%5d: %7.1f us %5.0f TF rc=%d S=%d | WGs %4d stages/WG %5.1f | per stage: wait %5.0f bar %5.0f issue %5.0f comp %5.0f"
               " | per WG: epi %6.0f total %7.0f (max %7llu) | start: prod init %5.0f (args %5.0f, index math %5.0f, offsets %5.0f) prime %5.0f first-land %5.0f; cons B0 at %6.0f\n",
               s.name, (long)s.M, s.N, s.K, us, fl / us / 1e6, rc, (int)(work_bytes / (s.M * s.N * 4)), n, stg, sum[0] / sum[5], sum[1] / sum[5], sum[2] / sum[5],
               sum[3] / sum[5], sum[4] / std::max(n, 1), sum[6] / std::max(n, 1), maxtot, sum[8] / std::max(n, 1), sum[12] / std::max(n, 1),
               sum[13] / std::max(n, 1), sum[14] / std::max(n, 1), sum[9] / std::max(n, 1),
               sum[10] / std::max(n, 1), sum[11] / std::max(n, 1));
        hipFree(a); hipFree(b); hipFree(c); hipFree(c2); hipFree(aux); hipFree(bias);
        if (work) hipFree(work);
    }
    return 0;
}
