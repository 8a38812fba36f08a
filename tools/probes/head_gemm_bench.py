"""The exact-f32 few-row GEMM of the projection heads (clv_sgemm_strided, forward form x W^T + b) at the step's shapes:
device time per launch by hipGraph replay.   python tools/probes/head_gemm_bench.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops


def graph_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * n) * 1e3


flush = torch.empty(512 << 20, device='cuda', dtype=torch.uint8)
for (M, N, K) in [(16, 1536, 768), (16, 768, 1536), (16, 768, 768), (32, 768, 768), (8, 1536, 768)]:
    x = torch.randn(M, K, device='cuda')
    ws = [torch.randn(N, K, device='cuda') * 0.03 for _ in range(20)]        # 20 weights: no launch finds its weight in L2
    b = torch.randn(N, device='cuda')
    y = torch.empty(M, N, device='cuda')
    it = [0]

    def fwd():
        w = ws[it[0] % len(ws)]
        it[0] += 1
        ops._sgemm_strided(x, (K, 1), w, (K, 1), b, y, M, N, K, False)
    t = graph_time(fwd)
    ref = x.double() @ ws[(it[0] - 1) % len(ws)].double().t() + b.double()
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    print(f'M={M:3d} N={N:5d} K={K:5d}: {t:6.1f} us  ({N * K * 4 / t / 1e6:6.2f} TB/s of weight)  rel err {err:.1e}', flush=True)
