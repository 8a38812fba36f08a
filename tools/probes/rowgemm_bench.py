import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from clover_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, K, name) in [(200704, 288, 96, 'qkv s0'), (200704, 96, 96, 'proj s0'), (200704, 384, 96, 'fc1 s0'), (200704, 96, 384, 'fc2 s0'),
                        (200704, 96, 288, 'qkv dgrad s0'), (50176, 576, 192, 'qkv s1'), (50176, 768, 192, 'fc1 s1'), (50176, 192, 768, 'fc2 s1'),
                        (12544, 1152, 384, 'qkv s2')]:
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.1).to(torch.bfloat16)
    b = torch.randn(N, device='cuda'); bb = b.to(torch.bfloat16)
    t_lib = timeit(lambda: F.linear(x, w, bb))
    t_own = timeit(lambda: ops.rowgemm(x, w, b)) if ops.rowgemm_supported(N, K) else float('nan')
    t_std = timeit(lambda: ops.rowgemm(x, w, b, standardise=True)) if ops.rowgemm_supported(N, K, True) else float('nan')
    gb = M * (K + N) * 2 / 1e9
    print(f'{name:14s} M={M} N={N} K={K}: hipBLASLt {t_lib:7.1f} us ({gb / t_lib * 1e3:5.2f} TB/s)  rowgemm {t_own:7.1f} us ({gb / t_own * 1e3:5.2f} TB/s)  +LN {t_std:7.1f} us')
