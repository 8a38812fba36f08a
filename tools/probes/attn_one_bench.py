"""clv_attn_bwd (all stages of the call: dQ / dK / dV / table gradient) at the step's four window-attention shapes (Swin-T, 16
clips x 8 frames), device time per call from a hipGraph of 10 calls (us).  CLV_ATTN_BWD_ONE=0|1, CLOVER_LIB_PATH=<ablation build>."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from clover_amd import ops, _lib
from clover_amd._lib import ClvAttnGeom
from clover_amd.backbones.swin_transformer_3d import window_geometry

L = _lib.lib()


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


out = []
SHAPES = {'T8': ([(16, 4, 56, 56, 96, 3), (16, 4, 28, 28, 192, 6), (16, 4, 14, 14, 384, 12), (16, 4, 7, 7, 768, 24)], (2, 2, 6, 2)),
          'B16': ([(8, 8, 56, 56, 128, 4), (8, 8, 28, 28, 256, 8), (8, 8, 14, 14, 512, 16), (8, 8, 7, 7, 1024, 32)], (2, 2, 18, 2))}
shapes, depths = SHAPES[os.environ.get('SHAPES', 'T8')]       # Swin-T 8 frames x 16 clips (196-token windows) / Swin-B 16 frames x 8 clips (392)
for (B, D, H, W, Cc, nH) in shapes:
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

    def run():
        rc = L.clv_attn_bwd(C.c_void_p(p), C.c_void_p(p + 2 * Cc), C.c_void_p(p + 4 * Cc), P(o), P(do), P(lse), P(table), P(r), None,
                            C.c_void_p(d), C.c_void_p(d + 2 * Cc), C.c_void_p(d + 4 * Cc), P(dtab), P(dsum), P(work), None, 0,
                            C.byref(g), st())
        assert rc == 0, rc
    out.append(graph_time(run))
    if os.environ.get('ONE_TRACE'):          # a -DONE_TRACE build leaves s_memtime deltas of every workgroup in dsum (16 floats each)
        dsum.zero_()
        run()
        torch.cuda.synchronize()
        nwg = g.groups * nH
        t = dsum[:nwg * 16].view(nwg, 16).double()
        m = t.mean(0).tolist()
        print(f'  {nwg} WGs, cycles from kernel entry (mean): rows {m[0]:.0f} | staged {m[1]:.0f} | B0 {m[2]:.0f} | loop end {m[3]:.0f} | '
              f'stores issued {m[4]:.0f} || service: loop end {m[5]:.0f}, last dQ {m[6]:.0f} || cycles waiting at the chunk barriers: '
              f'wave 0 {m[7]:.0f}, last compute wave {m[9]:.0f}, service wave {m[8]:.0f}')
print(os.environ.get('CLOVER_LIB_PATH', 'default').split('/')[-1], 'ONE=' + os.environ.get('CLV_ATTN_BWD_ONE', '1'),
      ' '.join(f'{t:7.1f}' for t in out), f'| step total {sum(d * t for d, t in zip(depths, out)):7.1f} us')
