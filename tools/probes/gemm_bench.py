"""LDS-tiled NT GEMM (clv_gemm_nt) vs the tuned library GEMM on the step's mid-size Linear shapes: forward (bias),
fc1 forward (+GELU vs library + gelu kernel), dgrad with the GELU-backward epilogue vs library + kernel."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from clover_amd import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lib_gemm_tuning import enable_tuned_gemms
enable_tuned_gemms()
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
SH = [(50176, 576, 192, 'qkv s1'), (50176, 192, 192, 'proj s1'), (50176, 768, 192, 'fc1 s1'), (50176, 192, 768, 'fc2 s1'),
      (50176, 192, 384, 'merge s1'),
      (12544, 1152, 384, 'qkv s2'), (12544, 384, 384, 'proj s2'), (12544, 1536, 384, 'fc1 s2'), (12544, 384, 1536, 'fc2 s2'),
      (12544, 384, 768, 'merge s2'),
      (3136, 2304, 768, 'qkv s3'), (3136, 768, 768, 'proj s3'), (3136, 3072, 768, 'fc1 s3'), (3136, 768, 3072, 'fc2 s3'),
      (3648, 2304, 768, 'qkv fu'), (3648, 768, 768, 'out fu'), (3648, 3072, 768, 'fc1 fu'), (3648, 768, 3072, 'fc2 fu'),
      (512, 2304, 768, 'qkv bert'), (512, 3072, 768, 'fc1 bert'), (512, 768, 3072, 'fc2 bert')]
only = sys.argv[1] if len(sys.argv) > 1 else ''
for (M, N, K, name) in SH:
    if only and only not in name: continue
    x = torch.randn(M, K, device='cuda').to(ops.BF16); w = (torch.randn(N, K, device='cuda') * 0.05).to(ops.BF16)
    b = torch.randn(N, device='cuda'); bb = b.to(ops.BF16)
    pre = torch.randn(M, N, device='cuda').to(ops.BF16)
    t_lib = timeit(lambda: F.linear(x, w, bb))
    t_own = timeit(lambda: ops.gemm_nt(x, w, b, epilogue=1))
    def lib_gelu():
        h = F.linear(x, w, bb); return ops.gelu(h)
    t_libg = timeit(lib_gelu)
    t_owng = timeit(lambda: ops.gemm_nt(x, w, b, epilogue=2))
    def lib_dgelu():
        d = F.linear(x, w); o = torch.empty_like(d)
        ops.check(ops._lib.lib().clv_gelu_bwd(ops._ptr(d), ops._ptr(pre), ops._ptr(o), d.numel(), 0, ops._stream()), 'g'); return o
    t_libd = timeit(lib_dgelu)
    t_ownd = timeit(lambda: ops.gemm_nt(x, w, aux=pre, epilogue=3))
    by = (M * K + N * K + M * N) * 2; fl = 2 * M * N * K
    ideal = max(by / 6.3e12, fl / 2.5e15) * 1e6
    print(f'{name:9s} M={M:6d} N={N:5d} K={K:5d}: bias lib {t_lib:6.1f} own {t_own:6.1f} | +gelu lib {t_libg:6.1f} own {t_owng:6.1f} | '
          f'dgelu lib {t_libd:6.1f} own {t_ownd:6.1f} | ideal {ideal:5.1f} us  ({fl / t_own / 1e6:6.0f} TF own, {fl / t_lib / 1e6:6.0f} TF lib)', flush=True)
