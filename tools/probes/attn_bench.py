"""Window-attention forward + backward (table gradient included) at the step's four stage shapes (Swin-T, 16 clips x 8 frames),
device-side time per call by hipGraph replay (us): fwd | fwd + bwd.   CLOVER_LIB_PATH=... python tools/probes/attn_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops
from clover_amd.backbones.swin_transformer_3d import window_geometry


def graph_time(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(3): g.replay()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / (3 * n) * 1e3


for (B, D, H, W, C, nH) in [(16, 4, 56, 56, 96, 3), (16, 4, 28, 28, 192, 6), (16, 4, 14, 14, 384, 12), (16, 4, 7, 7, 768, 24)]:
    for shifted in (False, True):
        ws, ss, rid = window_geometry((D, H, W), (8, 7, 7), (4, 3, 3) if shifted else (0, 0, 0), 'cuda')
        qkv = torch.randn(B, D, H, W, 3 * C, device='cuda').to(torch.bfloat16).requires_grad_()
        table = (torch.randn(15 * 13 * 13, nH, device='cuda') * 0.5).requires_grad_()
        table._clv_grad, table._clv_ready = torch.zeros_like(table), (lambda: None)
        do = torch.randn(B, D, H, W, C, device='cuda').to(torch.bfloat16)
        r = rid if any(s > 0 for s in ss) else None

        def fwd():
            with torch.no_grad():
                return ops.window_attention(qkv, table, r, ws, ss, nH, table_window=(8, 7, 7))

        def both():
            o = ops.window_attention(qkv, table, r, ws, ss, nH, table_window=(8, 7, 7))
            o.backward(do)
            qkv.grad = None
        tf, tb = graph_time(fwd), graph_time(both)
        print(f'stage C={C:4d} nH={nH:2d} shifted={int(shifted)}: fwd {tf:6.1f} | fwd+bwd {tb:6.1f} | bwd {tb - tf:6.1f}', flush=True)


# ---- the backward's kernels one by one (clv_attn_bwd stage masks: 1 = dQ (+ dS scratch), 2 = table gradient, 4 = dK / dV)
import ctypes as C
from clover_amd import _lib
from clover_amd._lib import ClvAttnGeom
L = _lib.lib()
print('backward kernels (us): dQ | table gradient (sum + gather) | dK dV')
for (B, D, H, W, Cc, nH) in [(16, 4, 56, 56, 96, 3), (16, 4, 28, 28, 192, 6), (16, 4, 14, 14, 384, 12), (16, 4, 7, 7, 768, 24)]:
    ws, ss, rid = window_geometry((D, H, W), (8, 7, 7), (4, 3, 3), 'cuda')
    N = ws[0] * ws[1] * ws[2]
    nW = (D // ws[0]) * (H // ws[1]) * (W // ws[2])
    hd = Cc // nH
    g = ClvAttnGeom(mode=1, groups=B * nW, N=N, nH=nH, hd=hd, D=D, H=H, W=W, wd=ws[0], wh=ws[1], ww=ws[2], sd=ss[0], sh=ss[1],
                    sw=ss[2], ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, bwd=8, bwh=7, bww=7, scale=hd ** -0.5, dropout_p=0.0)
    qkv = torch.randn(B, D, H, W, 3 * Cc, device='cuda').to(torch.bfloat16)
    table = torch.randn(15 * 13 * 13, nH, device='cuda') * 0.5
    o = torch.empty(B, D, H, W, Cc, device='cuda', dtype=torch.bfloat16)
    lse = torch.empty(g.groups * nH * N, device='cuda')
    r = rid if any(s > 0 for s in ss) else None
    p = qkv.data_ptr()
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    L.clv_attn_fwd(C.c_void_p(p), C.c_void_p(p + 2 * Cc), C.c_void_p(p + 4 * Cc), P(o), P(lse), P(table), P(r), None, None, C.byref(g), st())
    do = torch.randn_like(o)
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty_like(lse)
    dtab = torch.zeros_like(table)
    work = torch.empty(L.clv_attn_bwd_work_bytes(C.byref(g)), device='cuda', dtype=torch.uint8)
    g.dbias_index = ops._dbias_index(g, qkv.device)
    d = dqkv.data_ptr()

    def run(mask):
        rc = L.clv_attn_bwd(C.c_void_p(p), C.c_void_p(p + 2 * Cc), C.c_void_p(p + 4 * Cc), P(o), P(do), P(lse), P(table), P(r), None,
                            C.c_void_p(d), C.c_void_p(d + 2 * Cc), C.c_void_p(d + 4 * Cc), P(dtab), P(dsum), P(work), None, mask,
                            C.byref(g), st())
        assert rc == 0, rc
    t1, t2, t4, t7 = (graph_time(lambda m=m: run(m)) for m in (1, 2, 4, 7))
    print(f'stage C={Cc:4d}: dQ {t1:6.1f} | table {t2:6.1f} | dK dV {t4:6.1f} | all in one call {t7:6.1f}', flush=True)
