"""Does running the weight-gradient kernel of a Linear on a second stream, concurrently with the input-gradient GEMM
chain, pay inside a hipGraph?  Chain of L layers (stage-2 shapes): dgrad (library GEMM) -> next layer's dgrad ...;
wgrad of each layer depends only on that layer's dy.  Variants: serial (one stream), forked (wgrads on a side stream,
one join at the end), both as captured graphs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 12544
dims = [(384, 1536), (1536, 384)] * 6                      # (N out-features of this layer's dy, K in-features)
dev = 'cuda'
W = [(torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16) for n, k in dims]
X = [torch.randn(M, k, device=dev).to(torch.bfloat16) for n, k in dims]
DW = [torch.zeros(n, k, device=dev) for n, k in dims]
DB = [torch.zeros(n, device=dev) for n, k in dims]
dy0 = torch.randn(M, dims[0][0], device=dev).to(torch.bfloat16)
side = torch.cuda.Stream()

def run(fork):
    dy = dy0
    main = torch.cuda.current_stream()
    for i, (n, k) in enumerate(dims):
        if fork:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.linear_wgrad(dy, X[i], True, DW[i], DB[i])
        else:
            ops.linear_wgrad(dy, X[i], True, DW[i], DB[i])
        dy = torch.mm(dy, W[i])                             # [M, n] x [n, k] -> dx = next layer's dy
    if fork:
        main.wait_stream(side)
    return dy

def bench(fork, graph):
    for _ in range(3): run(fork)
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run(fork)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            run(fork)
        fn = g.replay
    else:
        fn = lambda: run(fork)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3

for graph in (False, True):
    for fork in (False, True):
        print(f'M={M} graph={graph} fork={fork}: {bench(fork, graph):8.1f} us per {len(dims)}-layer chain', flush=True)
