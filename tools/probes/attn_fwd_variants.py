"""Window-attention forward through the C entry at the step's four stage shapes (Swin-T, 16 clips x 8 frames): device time per
launch (hipGraph replay, us) with / without the relative-position table and the shift mask — what the bias gathers and the
region mask cost inside attn_fwd_kernel<32,13>.   python tools/probes/attn_fwd_variants.py"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import _lib, ops
from clover_amd._lib import ClvAttnGeom
from clover_amd.backbones.swin_transformer_3d import window_geometry

L = _lib.lib()
HALF = ops.BF16


def graph_time(fn, n=10):
    for _ in range(3):
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


P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
print('fwd us: plain+table | plain, no table | shifted+table | shifted, no table')
for (B, D, H, W, Cc, nH) in [(16, 4, 56, 56, 96, 3), (16, 4, 28, 28, 192, 6), (16, 4, 14, 14, 384, 12), (16, 4, 7, 7, 768, 24)]:
    row = []
    for shifted in (False, True):
        ws, ss, rid = window_geometry((D, H, W), (8, 7, 7), (4, 3, 3) if shifted else (0, 0, 0), 'cuda')
        N = ws[0] * ws[1] * ws[2]
        nW = (D // ws[0]) * (H // ws[1]) * (W // ws[2])
        hd = Cc // nH
        g = ClvAttnGeom(mode=1, groups=B * nW, N=N, nH=nH, hd=hd, D=D, H=H, W=W, wd=ws[0], wh=ws[1], ww=ws[2], sd=ss[0],
                        sh=ss[1], sw=ss[2], ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, bwd=8, bwh=7, bww=7, scale=hd ** -0.5,
                        dropout_p=0.0)
        qkv = torch.randn(B, D, H, W, 3 * Cc, device='cuda').to(HALF)
        table = torch.randn(15 * 13 * 13, nH, device='cuda') * 0.5
        o = torch.empty(B, D, H, W, Cc, device='cuda', dtype=HALF)
        lse = torch.empty(g.groups * nH * N, device='cuda')
        r = rid if any(s > 0 for s in ss) else None
        p = qkv.data_ptr()
        for tab in (table, None):
            def fwd():
                rc = L.clv_attn_fwd(C.c_void_p(p), C.c_void_p(p + 2 * Cc), C.c_void_p(p + 4 * Cc), P(o), P(lse), P(tab), P(r),
                                    None, None, C.byref(g), st())
                assert rc == 0, rc
            row.append(graph_time(fwd))
    print(f'C={Cc:4d} nH={nH:2d} windows={B * nW:5d}: ' + ' | '.join(f'{t:6.1f}' for t in row), flush=True)
