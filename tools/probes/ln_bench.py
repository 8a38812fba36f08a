"""LayerNorm forward / backward (ops.layer_norm) at the video tower's shapes, cold operands (a rotation of buffers larger than
the Infinity Cache): device time per launch by hipGraph replay and the bandwidth on the bytes it must move.
    python tools/probes/ln_bench.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops


def graph_time(fn, n):
    for _ in range(2):
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


for (rows, C) in [(200704, 96), (50176, 192), (12544, 384), (3136, 768), (50176, 384), (12544, 768)]:
    nb = max(2, int(600e6 // (rows * C * 2)) + 1)
    xs = [torch.randn(rows, C, device='cuda').to(ops.BF16).requires_grad_() for _ in range(nb)]
    rs = [torch.randn(rows, C, device='cuda').to(ops.BF16) for _ in range(nb)]
    dys = [torch.randn(rows, C, device='cuda').to(ops.BF16) for _ in range(nb)]
    w = torch.ones(C, device='cuda', requires_grad=True)
    b = torch.zeros(C, device='cuda', requires_grad=True)
    it = [0]

    def fwd():
        i = it[0] % nb
        it[0] += 1
        with torch.no_grad():
            return ops.layer_norm(xs[i], w, b, residual=rs[i], return_sum=True)

    def both():
        i = it[0] % nb
        it[0] += 1
        y, s = ops.layer_norm(xs[i], w, b, residual=rs[i], return_sum=True)
        torch.autograd.backward([y, s], [dys[i], dys[(i + 1) % nb]])
        xs[i].grad = None
    n = 2 * nb
    tf = graph_time(fwd, n)
    tb = graph_time(both, n) - tf
    byf, byb = rows * C * 2 * 4, rows * C * 2 * 4              # fwd: x, res in; y, sum out.  bwd: dy, dsum, sum in; dx out
    print(f'rows {rows:6d} C {C:4d}: fwd {tf:6.1f} us ({byf / tf / 1e6:5.2f} TB/s)   bwd {tb:6.1f} us ({byb / tb / 1e6:5.2f} TB/s)', flush=True)
    del xs, rs, dys
