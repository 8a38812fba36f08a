"""AdamW kernel alone on a 160 M-float slab: time per launch and effective HBM bandwidth (30 B per parameter)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops
n = 160_000_000
dev = 'cuda'
p = torch.randn(n, device=dev); g = torch.randn(n, device=dev) * 1e-3
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
sh = torch.zeros(n, device=dev, dtype=torch.bfloat16)
ss = torch.zeros(1, device=dev)
ops.sumsq_accumulate(g, ss)
for _ in range(3):
    ops.adamw_step(p, g, m, v, sh, ss, 1e-4, 0.9, 0.98, 1e-8, 0.01, 1, 15.0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
N = 20
for i in range(N):
    ops.adamw_step(p, g, m, v, sh, ss, 1e-4, 0.9, 0.98, 1e-8, 0.01, 2 + i, 15.0)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / N
print(f'adamw {t * 1e3:.1f} us  {n * 30 / t / 1e9:.2f} TB/s  (V2={os.environ.get("CLOVER_ADAMW_V2", "0")})')
