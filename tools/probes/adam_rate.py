"""clv_adamw_step_dev on a 160 M-parameter slab: microseconds and HBM rate (30 B per parameter) for the CLV_ADAM_NT /
CLV_ADAM_GRID variants."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops
n = 160_000_000
p = torch.randn(n, device='cuda'); g = torch.randn(n, device='cuda') * 1e-3
m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda'); sh = torch.empty(n, device='cuda', dtype=torch.bfloat16)
st = ops.optim_state_new(p.device)
ss = torch.ones(1, device='cuda')
ops.optim_prep(ss, st, 0.9, 0.98, 0.0, 1.0)
def run(): ops.adamw_step_dev(p, g, m, v, sh, st, 1e-4, 0.9, 0.98, 1e-8, 0.01)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10 * 1e-3
print(f'NT={os.environ.get("CLV_ADAM_NT", "0")} GRID={os.environ.get("CLV_ADAM_GRID", "-")}: {t * 1e6:7.1f} us  {30 * n / t / 1e12:5.2f} TB/s')
