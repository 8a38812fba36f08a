"""Which torch ops (with shapes) launch the step's small elementwise kernels: one eager step under torch.profiler."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
import clover_amd
from clover_amd.engine import CloverEngine
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
model = clover_amd.build_model(bench.model_cfg('T', 8)).to(dev); model.train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(8, 8, 32, 1000).items()}
eng = CloverEngine(model, batch, lr=1e-5, weight_decay=0.005, grad_clip=15.0, max_iters=100000)
for _ in range(3): eng.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    eng.step(batch)
    torch.cuda.synchronize()
want = [a for a in sys.argv[1:] if not a.startswith('--')] or ['aten::add_', 'aten::add', 'aten::zeros', 'aten::zero_', 'aten::fill_', 'aten::copy_', 'aten::mul', 'aten::zeros_like', 'aten::sum', 'aten::to', 'aten::_to_copy', 'aten::contiguous']
cnt = collections.Counter(); tim = collections.Counter(); where = {}
for e in prof.events():
    if (e.name in want or '--all' in sys.argv) and e.device_time > 0 and e.name.startswith('aten::'):
        st = [s for s in (e.stack or []) if 'clover_amd' in s or 'bench.py' in s]
        key = (e.name, str(e.input_shapes)[:70], (st[0].split('/')[-1][:60] if st else '?'))
        cnt[key] += 1; tim[key] += e.device_time
order = sorted(tim.items(), key=lambda kv: -cnt[kv[0]]) if '--by-count' in sys.argv else sorted(tim.items(), key=lambda kv: -kv[1])
for k, t in order[:110]:
    print(f'{t:8.0f} us {cnt[k]:4d}x  {k[0]:16s} {k[1]:70s} {k[2]}')
