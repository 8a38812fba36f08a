"""clv_gemm_nt tile classes vs the tuned library GEMM on the long-contraction / few-tile Linear shapes of the step
(Swin stage 3, fusion encoder, text tower, long-K layers of stages 1-2).  Device-side time per launch by HIP events over
back-to-back launches (us).   python tools/probes/gemm_tiles.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from clover_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lib_gemm_tuning import enable_tuned_gemms
enable_tuned_gemms()


def timeit(fn, n=20):
    """Device-side time per launch: n launches captured in ONE hipGraph (no host launch gaps), replayed 3 times."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * n) * 1e3


SHORTK = [(50176, 576, 192, 'qkv s1'), (50176, 192, 192, 'proj s1'), (50176, 768, 192, 'fc1 s1'), (50176, 192, 576, 'dqkv s1'),
          (12544, 1152, 384, 'qkv s2'), (12544, 384, 384, 'proj s2'), (12544, 1536, 384, 'fc1 s2'), (12544, 384, 1152, 'dqkv s2')]
SH = [(50176, 192, 768, 'fc2 s1'), (12544, 384, 1536, 'fc2 s2'), (12544, 384, 768, 'merge s2'), (3136, 768, 1536, 'merge s3'),
      (3136, 2304, 768, 'qkv s3'), (3136, 768, 768, 'proj s3'), (3136, 3072, 768, 'fc1 s3'), (3136, 768, 3072, 'fc2 s3'),
      (3648, 2304, 768, 'qkv fu'), (3648, 768, 768, 'out fu'), (3648, 3072, 768, 'fc1 fu'), (3648, 768, 3072, 'fc2 fu'),
      (512, 2304, 768, 'qkv bert'), (512, 768, 768, 'out bert'), (512, 3072, 768, 'fc1 bert'), (512, 768, 3072, 'fc2 bert')]
TILES = [None, '128x128w4', '128x128w8r2']
if os.environ.get('SHAPES') == 'shortk':      # the short-contraction token-parallel layers of stages 1-2
    SH = SHORTK
for (M, N, K, name) in SH:
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda'); bb = b.to(torch.bfloat16)
    t_lib = timeit(lambda: F.linear(x, w, bb))
    res = []
    ref = F.linear(x, w, bb).float()
    for t in TILES:
        if t is None: os.environ.pop('CLV_GEMM_TILE', None)
        else: os.environ['CLV_GEMM_TILE'] = t
        try:
            y = ops.gemm_nt(x, w, b, epilogue=1)
            err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
            res.append((t or '128x128', timeit(lambda: ops.gemm_nt(x, w, b, epilogue=1)), err))
        except RuntimeError as e:
            res.append((t or '128x128', float('nan'), -1))
    os.environ.pop('CLV_GEMM_TILE', None)
    fl = 2 * M * N * K
    print(f'{name:9s} M={M:6d} N={N:5d} K={K:5d}: lib {t_lib:6.1f} ({fl / t_lib / 1e6:5.0f} TF) | ' +
          ' | '.join(f'{t} {v:6.1f}{"" if e < 2e-2 else " BAD%.3f" % e}' for t, v, e in res), flush=True)
