"""LDS-tiled NT GEMM (clv_gemm_nt) vs the library on the step's mid-size Linear shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from clover_amd import _lib
from clover_amd.utils.gemm_tuning import enable_tuned_gemms
enable_tuned_gemms()
L = _lib.lib()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
SH = [(50176, 576, 192, 'qkv s1'), (50176, 192, 192, 'proj s1'), (50176, 768, 192, 'fc1 s1'), (50176, 192, 768, 'fc2 s1'),
      (12544, 1152, 384, 'qkv s2'), (12544, 384, 384, 'proj s2'), (12544, 1536, 384, 'fc1 s2'), (12544, 384, 1536, 'fc2 s2'),
      (3136, 2304, 768, 'qkv s3'), (3136, 3072, 768, 'fc1 s3'), (3136, 768, 3072, 'fc2 s3'), (3648, 2304, 768, 'qkv fu'),
      (3648, 3072, 768, 'fc1 fu'), (3648, 768, 3072, 'fc2 fu'), (12544, 384, 768, 'merge s2'), (1000, 136, 96, 'ragged')]
st = torch.cuda.current_stream().cuda_stream
for (M, N, K, name) in SH:
    x = torch.randn(M, K, device='cuda').to(torch.bfloat16); w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda'); bb = b.to(torch.bfloat16)
    y = torch.empty(M, N, device='cuda', dtype=torch.bfloat16); pre = torch.empty_like(y)
    def own(epi=0):
        rc = L.clv_gemm_nt(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), pre.data_ptr(), M, N, K, K, N, epi, st)
        assert rc == 0, rc
    own(0)
    ref = F.linear(x.float(), w.float(), b)
    err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    own(1)
    errg = ((y.float() - F.gelu(ref)).abs().max() / ref.abs().max()).item()
    errp = ((pre.float() - ref).abs().max() / ref.abs().max()).item()
    t_lib = timeit(lambda: F.linear(x, w, bb)); t_own = timeit(lambda: own(0)); t_g = timeit(lambda: own(1))
    by = (M * K + N * K + M * N) * 2; fl = 2 * M * N * K
    ideal = max(by / 6.3e12, fl / 2.5e15) * 1e6
    print(f'{name:9s} M={M:6d} N={N:5d} K={K:5d}: lib {t_lib:6.1f} us  own {t_own:6.1f} us  own+gelu {t_g:6.1f} us  ideal {ideal:5.1f} us  err {err:.1e} {errg:.1e} {errp:.1e}')
