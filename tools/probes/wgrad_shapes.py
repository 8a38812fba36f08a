"""Each weight-gradient shape 12 times, a marker fill between shapes (so a kernel trace splits into one run per shape).
Prints the shape list; read durations with tools/probes/trace_summary.py <csv> wgrad / fold."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import _lib
L = _lib.lib()
SHAPES = [(200704, 384, 96, 'fc1 s0'), (200704, 96, 384, 'fc2 s0'), (50176, 576, 192, 'qkv s1'), (50176, 768, 192, 'fc1 s1'), (50176, 192, 768, 'fc2 s1'),
          (12544, 1152, 384, 'qkv s2'), (12544, 384, 384, 'proj s2'), (12544, 1536, 384, 'fc1 s2'), (12544, 384, 1536, 'fc2 s2'),
          (3136, 768, 768, 'proj s3'), (3136, 2304, 768, 'qkv s3'), (3136, 3072, 768, 'fc1 s3'), (3136, 768, 3072, 'fc2 s3'),
          (3648, 2304, 768, 'qkv fu'), (3648, 3072, 768, 'fc1 fu'), (512, 768, 768, 'bert proj'), (512, 3072, 768, 'bert fc1')]
marker = torch.zeros(1 << 20, device='cuda')
for (M, N, K, name) in SHAPES:
    dy = torch.randn(M, N, device='cuda').to(torch.bfloat16); x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    dw = torch.zeros(N, K, device='cuda'); db = torch.zeros(N, device='cuda')
    work = torch.empty(L.clv_linear_wgrad_work_floats(M, N, K), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    marker.fill_(1.0)
    for _ in range(12):
        assert L.clv_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), work.data_ptr(), M, N, K, N, K, None, None, 3, st) == 0
    torch.cuda.synchronize()
print(' | '.join(s[3] for s in SHAPES))
