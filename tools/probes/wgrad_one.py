"""One weight-gradient shape in a loop (for rocprofv3 --pmc / --kernel-trace): wgrad_one.py M N K"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import _lib
L = _lib.lib()
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (12544, 1536, 384)
dy = torch.randn(M, N, device='cuda').to(torch.bfloat16); x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
dw = torch.zeros(N, K, device='cuda'); db = torch.zeros(N, device='cuda')
work = torch.empty(L.clv_linear_wgrad_work_floats(M, N, K), device='cuda')
st = torch.cuda.current_stream().cuda_stream
for _ in range(20):
    assert L.clv_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), work.data_ptr(), M, N, K, N, K, None, None, 3, st) == 0
torch.cuda.synchronize()
