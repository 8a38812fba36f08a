"""Do the two towers really overlap inside the replayed hipGraphs?  rocprofv3's kernel trace perturbs exactly that (in
its trace of a step the text tower's forward runs BEFORE the video tower's, not beside it), so this probe takes
device-side time stamps instead: single-lane kernels (tools/probes/stamp.hip) that store the 100 MHz real-time counter,
launched on the stream the surrounding code runs on — captured into the graphs with the step's own kernels — at
    the entry / exit of  encode,  text tower,  video tower,  fusion encoder           (forward)
    the arrival of the gradient at the towers' outputs and at their first layers      (backward, tensor hooks)
    either side of the optimizer.
Prints the mean position of every stamp in the step (us from the entry of encode) over the last steps.

    hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probes/bin/libstamp.so tools/probes/stamp.hip
    gpurun -- 'python tools/probes/graph_overlap.py'          (CLOVER_* switches apply as in bench.py)"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench, clover_amd
from clover_amd.engine import CloverEngine

lib = ctypes.CDLL(os.path.join(ROOT, 'tools/probes/bin/libstamp.so'))
lib.probe_stamp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda', 0)
NAMES = ['encode in', 'text fwd in', 'text fwd out', 'video fwd in', 'video fwd out', 'fusion fwd in', 'fusion fwd out',
         'encode out', 'fusion bwd: grad at its output', 'video bwd: grad at its output', 'video bwd: grad at patch embed',
         'text bwd: grad at its output', 'text bwd: grad at the embeddings', 'optimizer in', 'optimizer out',
         'before forward graph', 'after forward graph', 'before loss graph', 'after loss graph', 'before backward graph',
         'after backward graph']
buf = torch.zeros(len(NAMES), dtype=torch.int64, device=dev)


def stamp(name):
    rc = lib.probe_stamp(buf.data_ptr(), NAMES.index(name), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def around(mod, attr, before, after, out_hook=None, pick=lambda o: o):
    fn = getattr(mod, attr)

    def wrapped(*a, **k):
        if before:
            stamp(before)
        out = fn(*a, **k)
        if after:
            stamp(after)
        t = pick(out)
        if out_hook and torch.is_tensor(t) and t.requires_grad:
            t.register_hook(lambda g, n=out_hook: stamp(n))
        return out
    setattr(mod, attr, wrapped)


torch.manual_seed(1234)
model = clover_amd.build_model(bench.model_cfg('T', 8)).to(dev); model.train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(8, 8, 32, 1000).items()}
around(model, 'encode', 'encode in', 'encode out')
around(model.text_backbone, 'forward', 'text fwd in', 'text fwd out', 'text bwd: grad at its output',
       pick=lambda o: o['last_hidden_state'])
around(model.text_backbone.bert.embeddings, 'forward', None, None, 'text bwd: grad at the embeddings')
around(model.backbone, 'forward_both', 'video fwd in', 'video fwd out', 'video bwd: grad at its output')
around(model.backbone.patch_embed, 'tokens_stacked', None, None, 'video bwd: grad at patch embed')
around(model.multimodal_backbone, 'forward', 'fusion fwd in', 'fusion fwd out', 'fusion bwd: grad at its output',
       pick=lambda o: o['last_hidden_state'])
eng = CloverEngine(model, batch, lr=1e-5, weight_decay=0.005, grad_clip=15.0, max_iters=100000)
around(eng, 'optimizer_step', 'optimizer in', 'optimizer out')
eng.step(batch)
eng.capture(batch)


class Stamped:
    """A captured graph whose replay is bracketed by eager stamps on the launching stream."""

    def __init__(self, g, name):
        self.g, self.name = g, name

    def replay(self):
        stamp(f'before {self.name} graph')
        self.g.replay()
        stamp(f'after {self.name} graph')

    def __getattr__(self, k):
        return getattr(self.g, k)


eng.graph = Stamped(eng.graph, 'forward')
eng.graph_bwd = Stamped(eng.graph_bwd, 'backward')
if getattr(eng, 'graph_loss', None) is not None:
    eng.graph_loss = Stamped(eng.graph_loss, 'loss')
for _ in range(10):
    eng.step(batch)
torch.cuda.synchronize()
N = 30
acc = torch.zeros(len(NAMES), dtype=torch.float64)
seen = torch.zeros(len(NAMES))
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
steps = []
for _ in range(N):
    eng.step(batch)
    steps.append(buf.clone())
t1.record()
torch.cuda.synchronize()
for s in steps:
    s = s.cpu()
    ok = s > 0
    acc += torch.where(ok, (s - s[0]).double() / 100.0, torch.zeros((), dtype=torch.float64))
    seen += ok.float()
print(f'{t0.elapsed_time(t1) / N:.3f} ms / step with the stamps in ({N} steps, hipGraph replay)')
for i in sorted(range(len(NAMES)), key=lambda i: acc[i] / max(1.0, seen[i])):
    print(f'{acc[i] / max(1.0, seen[i]):9.1f} us  {NAMES[i]}' + ('' if seen[i] == N else f'   (seen {int(seen[i])} / {N})'))
