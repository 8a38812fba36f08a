"""Micro-benchmark of the window-attention kernels at the stage-0 shape of config 2 (2B = 16 clips)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import ops, _lib
if os.environ.get('PROBE_LIB'): _lib.LIB_PATH = os.path.abspath(os.environ['PROBE_LIB'])
from clover_amd.backbones.swin_transformer_3d import window_geometry
B, D, H, W, C, nH = 16, 4, 56, 56, 96, 3
if len(sys.argv) > 1 and sys.argv[1] == 's2':
    B, D, H, W, C, nH = 16, 4, 14, 14, 384, 12
if len(sys.argv) > 1 and sys.argv[1] == 'd8':            # 16-frame clips: windows of 8 x 7 x 7 = 392 tokens (Swin-B stage 0)
    B, D, H, W, C, nH = 16, 8, 56, 56, 128, 4
torch.manual_seed(0)
qkv = torch.randn(B, D, H, W, 3 * C, device='cuda').to(torch.bfloat16).requires_grad_()
table = (torch.randn(15 * 13 * 13, nH, device='cuda') * 0.5).requires_grad_()
SHIFT = (0, 0, 0) if os.environ.get('PROBE_NOSHIFT') else (4, 3, 3)
ws, ss, rid = window_geometry((D, H, W), (8, 7, 7), SHIFT, 'cuda')
if not any(SHIFT): rid = None
do = torch.randn(B, D, H, W, C, device='cuda').to(torch.bfloat16)
for it in range(5):
    o = ops.window_attention(qkv, table, rid, ws, ss, nH, table_window=(8, 7, 7))
    o.backward(do)
torch.cuda.synchronize()
filler = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
ops.PROF = {}
for _ in range(10):
    for _ in range(10): torch.mm(filler, filler)          # keep the host ahead so kernels run back-to-back
    o = ops.window_attention(qkv, None if os.environ.get('PROBE_NOBIAS') else table, rid, ws, ss, nH, table_window=(8, 7, 7))
    o.backward(do)
torch.cuda.synchronize()
prof, ops.PROF = ops.PROF, None
for k, evs in prof.items():
    ms = [a.elapsed_time(b) for a, b, _, _ in evs]
    print(f'{k:45s} {1e3 * sum(ms) / len(ms):8.1f} us')
