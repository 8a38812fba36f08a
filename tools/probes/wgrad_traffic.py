"""Per-stage anatomy of the grouped weight-gradient launch (clv_linear_wgrad_batch): the video tower's deferred problems
of ONE Swin stage at a time (and all together), each set launched REPS times in a fixed order, so that a
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` (or WRITE_SIZE) pass over this script can be split per set by dispatch order
(tools/probes/wgrad_traffic_sum.py).  Without the profiler it prints event-timed microseconds per set.
Run: gpurun -- 'python tools/probes/wgrad_traffic.py'"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops

SETS = {
    's3': [(3136, 768, 768), (3136, 2304, 768), (3136, 3072, 768), (3136, 768, 3072)] * 2,
    's2': [(12544, 384, 1536), (12544, 1536, 384), (12544, 384, 384), (12544, 1152, 384)] * 6 + [(12544, 384, 768)],
    's1': [(50176, 192, 768), (50176, 768, 192), (50176, 192, 192), (50176, 576, 192)] * 2 + [(50176, 192, 384)],
    's0': [(200704, 96, 384), (200704, 96, 96)] * 2,
    's0x': [(200704, 96, 384), (200704, 96, 96), (200704, 288, 96), (200704, 384, 96)] * 2,   # incl. the LN-fused pair
}
SETS['all'] = SETS['s3'] + SETS['s2'] + SETS['s1'] + SETS['s0']
ORDER = os.environ.get('SETS', 's0,s0x,s1,s2,s3,all').split(',')
REPS = int(os.environ.get('REPS', '3'))
TIMED = int(os.environ.get('TIMED', '10'))


def make(P):
    pend = []
    for (M, N, K) in P:
        dy = torch.randn(M, N, device='cuda').to(torch.bfloat16)
        x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
        pend.append((dy, x, torch.zeros(N, K, device='cuda'), torch.zeros(N, device='cuda'), M, N, K))
    return pend


def run(pend):
    folds = ops.flush_wgrads(pend)
    if os.environ.get('NOFOLD') != '1':
        ops.flush_folds(folds)


MARK = torch.arange(4096, device='cuda', dtype=torch.float32)
for name in ORDER:
    P = SETS[name]
    pend = make(P)
    fl = sum(2 * M * N * K for (M, N, K) in P)
    by = sum(2 * M * (N + K) for (M, N, K) in P)
    for _ in range(REPS):
        run(pend)
    MARK.cumsum(0)                                          # set separator in a kernel trace (a scan kernel nothing else launches)
    torch.cuda.synchronize()
    line = f'SET {name} problems {len(P)} reps {REPS} alg_bytes {by} flops {fl}'
    if TIMED:
        # device-side time: the launches replayed from a hipGraph (the host needs ~100 us per run to build the tables)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run(pend)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            run(pend)
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(TIMED):
            g.replay()
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / TIMED * 1e-3
        del g
        line += f' us {t * 1e6:.1f} alg_TBps {by / t / 1e12:.2f} TFLOPs {fl / t / 1e12:.1f}'
    print(line, flush=True)
    del pend
    torch.cuda.empty_cache()
