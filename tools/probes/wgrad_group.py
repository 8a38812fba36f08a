"""The video tower's deferred weight gradients as the grouped launches the step makes (clv_linear_wgrad_batch), timed
with events; CLV_WGRAD_GROUP_TARGET overrides the per-problem workgroup target for sweeps."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops
P = [(3136, 768, 768)] * 2
for _ in range(6):
    P += [(12544, 384, 1536), (12544, 1536, 384), (12544, 384, 384), (12544, 1152, 384)]
P += [(12544, 384, 768)]
for _ in range(2):
    P += [(50176, 192, 768), (50176, 768, 192), (50176, 192, 192), (50176, 576, 192)]
P += [(50176, 192, 384)]
for _ in range(2):
    P += [(200704, 96, 384), (200704, 96, 96)]
only = os.environ.get('ONLY', '')          # 's0' / 's1' / 's2' / 's3': one stage's problems only
if only:
    rows = {'s0': 200704, 's1': 50176, 's2': 12544, 's3': 3136}[only]
    P = [p for p in P if p[0] == rows]
order = os.environ.get('ORDER', '')
if order == 'bigfirst':
    P.sort(key=lambda p: -p[0])
pend = []
for (M, N, K) in P:
    dy = torch.randn(M, N, device='cuda').to(ops.BF16); x = torch.randn(M, K, device='cuda').to(ops.BF16)
    pend.append((dy, x, torch.zeros(N, K, device='cuda'), torch.zeros(N, device='cuda'), M, N, K))
def run():
    return ops.flush_wgrads(pend)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
s.record()
for _ in range(n): run()
e.record(); torch.cuda.synchronize()
fl = sum(2 * M * N * K for (M, N, K) in P); by = sum(2 * M * (N + K) for (M, N, K) in P)
t = s.elapsed_time(e) / n * 1e-3
print(f'{len(P)} problems: {t * 1e6:7.1f} us   {fl / t / 1e12:6.1f} TFLOP/s   {by / t / 1e12:5.2f} TB/s algorithmic', flush=True)
