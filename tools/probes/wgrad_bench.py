"""Per-shape timing of the split-M weight-grad kernels (stage masks 1 = partial kernel, 2 = fold) vs torch.mm."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from clover_amd import _lib
L = _lib.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
SHAPES = [(200704, 96, 96, 'pe/proj s0'), (200704, 288, 96, 'qkv s0'), (200704, 384, 96, 'fc1 s0'), (200704, 96, 384, 'fc2 s0'),
          (50176, 192, 384, 'merge s1'), (50176, 576, 192, 'qkv s1'), (50176, 192, 192, 'proj s1'), (50176, 768, 192, 'fc1 s1'), (50176, 192, 768, 'fc2 s1'),
          (12544, 384, 768, 'merge s2'), (12544, 1152, 384, 'qkv s2'), (12544, 384, 384, 'proj s2'), (12544, 1536, 384, 'fc1 s2'), (12544, 384, 1536, 'fc2 s2'),
          (3136, 768, 768, 'proj s3'), (3136, 2304, 768, 'qkv s3'), (3136, 3072, 768, 'fc1 s3'), (512, 768, 768, 'bert proj'), (512, 2304, 768, 'bert qkv'), (512, 3072, 768, 'bert fc1'), (512, 768, 3072, 'bert fc2'), (2120, 768, 768, 'fusion')]
tot = 0
for (M, N, K, name) in SHAPES:
    dy = torch.randn(M, N, device='cuda').to(torch.bfloat16); x = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    dw = torch.zeros(N, K, device='cuda'); db = torch.zeros(N, device='cuda')
    work = torch.empty(L.clv_linear_wgrad_work_floats(M, N, K), device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    def run(stage):
        rc = L.clv_linear_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr(), work.data_ptr(), M, N, K, N, K, None, None, stage, st)
        assert rc == 0
    t1 = timeit(lambda: run(1)); t2 = timeit(lambda: run(2))
    t_lib = timeit(lambda: torch.mm(dy.t(), x))
    dw.zero_(); run(3)
    ref = torch.mm(dy.float().t(), x.float())
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    gb = M * (K + N) * 2 / 1e9
    tot += t1 + t2
    print(f'{name:10s} M={M:6d} N={N:4d} K={K:4d}: partial {t1:6.1f} us  fold {t2:5.1f} us  ({gb / (t1 + t2) * 1e3:5.2f} TB/s alg; work {work.numel() * 4 / 1e6:5.1f} MB)  torch.mm {t_lib:7.1f} us  err {err:.1e}')
print('sum', tot)
