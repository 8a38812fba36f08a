"""Same problems through the 128 x 128 and the 256 x 256 tile class of the grouped weight-gradient launch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops
M, N, K, n = [int(a) for a in sys.argv[1:5]]
pend = []
for i in range(n):
    dy = torch.randn(M, N, device='cuda').to(torch.bfloat16); x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    pend.append((dy, x, torch.zeros(N, K, device='cuda'), torch.zeros(N, device='cuda'), M, N, K))
for _ in range(3): ops.flush_wgrads(pend)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): ops.flush_wgrads(pend)
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 20 * 1e-3
print(f'BIG={os.environ.get("CLV_WGRAD_BIG", "1")} {n} x ({M},{N},{K}): {t * 1e6:7.1f} us  {n * 2 * M * N * K / t / 1e12:6.1f} TFLOP/s  {n * 2 * M * (N + K) / t / 1e12:5.2f} TB/s', flush=True)
