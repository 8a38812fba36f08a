"""clv_gemm_nt variants vs the tuned library GEMM on every Linear shape of the step that runs on it: K rotation
(CLV_GEMM_ROT), K slices (CLV_GEMM_SPLITK), tile classes (CLV_GEMM_TILE).  Device-side time per launch (us): n launches
captured in one hipGraph, replayed.   python tools/probes/gemm_sweep.py [quick] [coldw]
coldw: every launch of the graph reads a DIFFERENT copy of the weight (copies total > 600 MB: more than the 256 MB
Infinity Cache), as in the step, where a layer's weight was last touched a step ago; the activation stays hot (its
producer has just written it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from clover_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lib_gemm_tuning import enable_tuned_gemms
enable_tuned_gemms()


COLDW = 'coldw' in sys.argv


def timeit(fn, n=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * n) * 1e3


SH = [(50176, 576, 192, 'qkv s1'), (50176, 192, 192, 'proj s1'), (50176, 768, 192, 'fc1 s1'), (50176, 192, 768, 'fc2 s1'),
      (50176, 192, 576, 'dqkv s1'), (12544, 1152, 384, 'qkv s2'), (12544, 384, 384, 'proj s2'), (12544, 1536, 384, 'fc1 s2'),
      (12544, 384, 1536, 'fc2 s2'), (12544, 384, 1152, 'dqkv s2'), (12544, 384, 768, 'merge s2'), (3136, 768, 1536, 'merge s3'),
      (3136, 2304, 768, 'qkv s3'), (3136, 768, 768, 'proj s3'), (3136, 3072, 768, 'fc1 s3'), (3136, 768, 3072, 'fc2 s3'),
      (3136, 768, 2304, 'dqkv s3'), (3648, 2304, 768, 'qkv fu'), (3648, 768, 768, 'out fu'), (3648, 3072, 768, 'fc1 fu'),
      (3648, 768, 3072, 'fc2 fu'), (512, 2304, 768, 'qkv bert'), (512, 768, 768, 'out bert'), (512, 3072, 768, 'fc1 bert'),
      (512, 768, 3072, 'fc2 bert'), (512, 768, 2304, 'dqkv bert')]
# (label, env)
VARIANTS = [('base', dict(CLV_GEMM_ROT='0', CLV_GEMM_SPLITK='1')),
            ('rot', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='1')),
            ('rot+auto', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='0')),
            ('rot+s2', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='2')),
            ('rot+s4', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='4')),
            ('rot+s8', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='8')),
            ('rot 64x128', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='1', CLV_GEMM_TILE='64x128w4')),
            ('rot 128w8r2', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='1', CLV_GEMM_TILE='128x128w8r2')),
            ('rot 128w4', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='1', CLV_GEMM_TILE='128x128w4')),
            ('rot 256x128', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='1', CLV_GEMM_TILE='256x128')),
            ('rot 128w8r4', dict(CLV_GEMM_ROT='1', CLV_GEMM_SPLITK='1', CLV_GEMM_TILE='128x128w8'))]
KEYS = ('CLV_GEMM_ROT', 'CLV_GEMM_SPLITK', 'CLV_GEMM_TILE')
if 't192' in sys.argv:     # the default planner against the 128 x 192 tile class (N = 192 / 384 / 576 / 1152 layers)
    VARIANTS = [('planner', dict()), ('128x192w8', dict(CLV_GEMM_TILE='128x192w8')), ('64x192w4', dict(CLV_GEMM_TILE='64x192w4'))]
if 'quick' in sys.argv:
    VARIANTS = [VARIANTS[0], VARIANTS[1], VARIANTS[6], VARIANTS[7], VARIANTS[8]]
print('shape'.ljust(34) + 'lib'.rjust(7) + ''.join(v[0].rjust(12) for v in VARIANTS), flush=True)
for (M, N, K, name) in SH:
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda'); bb = b.to(torch.bfloat16)
    nw = max(20, min(160, (700 << 20) // (N * K * 2) + 1)) if COLDW else 1      # one graph launch per copy
    ws = [w] + [w.clone() for _ in range(nw - 1)]
    t_lib = timeit(lambda i: F.linear(x, ws[i % nw], bb), n=max(20, nw))
    ref = F.linear(x, w, bb).float()
    cells = []
    for label, env in VARIANTS:
        for k in KEYS: os.environ.pop(k, None)
        os.environ.update(env)
        try:
            y = ops.gemm_nt(x, w, b, epilogue=1)
            err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
            t = timeit(lambda i: ops.gemm_nt(x, ws[i % nw], b, epilogue=1), n=max(20, nw))
            cells.append(f'{t:7.1f}' + ('*' if t <= t_lib else ' ') + ('' if err < 2e-2 else ' BAD%.2f' % err))
        except RuntimeError:
            cells.append('    n/a ')
    for k in KEYS: os.environ.pop(k, None)
    print(f'{name:9s} {M:6d}x{N:5d}x{K:5d} ' + f'{t_lib:7.1f}' + ''.join(c.rjust(12) for c in cells), flush=True)
