"""Device-side time of the one-kernel MLP (clv_mlp_fused_fwd / _bwd) against the round-5 kernel sequence at a VideoSwin-T
stage-0 shape (M = 200 704 tokens, C = 96, hidden 384): events around 20 back-to-back launches behind a filler GEMM.
    python tools/probes/mlp_fused_bench.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from clover_amd import ops  # noqa: E402

BF = torch.bfloat16
DEV = 'cuda'


def timed(fn, n=20):
    filler = torch.randn(4096, 4096, device=DEV, dtype=BF)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for _ in range(10):
        torch.mm(filler, filler)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 200704
    C, Hd = 96, 384
    g = torch.Generator(device='cpu').manual_seed(1)
    a = torch.randn(M, C, generator=g).to(BF).to(DEV)
    r = torch.randn(M, C, generator=g).to(BF).to(DEV)
    w1f = (0.1 * torch.randn(Hd, C, generator=g)).to(BF).to(DEV)
    b1f = (0.1 * torch.randn(Hd, generator=g)).to(DEV)
    w2 = (0.05 * torch.randn(C, Hd, generator=g)).to(BF).to(DEV)
    b2 = (0.1 * torch.randn(C, generator=g)).to(DEV)
    w2t = w2.t().contiguous()
    w1t = w1f.t().contiguous()
    do = torch.randn(M, C, generator=g).to(BF).to(DEV)
    ds = torch.randn(M, C, generator=g).to(BF).to(DEV)
    sc = torch.full((16,), 1.1, device=DEV)
    rps = M // 16
    o = ops.mlp_fused_fwd(a, r, w1f, b1f, w2, b2, 1e-5, sc, rps)
    t_f = timed(lambda: ops.mlp_fused_fwd(a, r, w1f, b1f, w2, b2, 1e-5, sc, rps))
    t_b = timed(lambda: ops.mlp_fused_bwd(o['sum'], o['mean'], o['rstd'], do, ds, w1f, b1f, w2t, sc, rps, want_dres=True))
    # round-5 sequence: LN + fc1 + GELU (rowgemm), fc2 (gemm_nt); backward: fc2 dgrad + GELU', fc1 dgrad, LN backward
    o1 = ops.rowgemm(a, w1f, b1f, res=r, standardise=True, epilogue=1, want_xhat=True, xscale=sc, rows_per_sample=rps)
    t_f1 = timed(lambda: ops.rowgemm(a, w1f, b1f, res=r, standardise=True, epilogue=1, want_xhat=True, xscale=sc, rows_per_sample=rps))
    t_f2 = timed(lambda: ops.gemm_nt(o1['y'], w2, b2, epilogue=ops.GEMM_EPI_BIAS))
    dpre = ops.rowgemm(do, w2t, None, epilogue=2, pre_in=o1['pre'])['y']
    t_b1 = timed(lambda: ops.rowgemm(do, w2t, None, epilogue=2, pre_in=o1['pre']))
    t_b2 = timed(lambda: ops.rowgemm(dpre, w1t, None))
    dxh = ops.rowgemm(dpre, w1t, None)['y']
    gam = torch.ones(C, device=DEV)
    t_b3 = timed(lambda: ops._ln_bwd_noaffine(dxh, o1['sum'], o1['mean'], o1['rstd'], ds, gamma=gam, xscale=sc, rows_per_sample=rps))
    nb_f = M * C * 2 * 4
    nb_b = M * C * 2 * 6 + 2 * M * Hd * 2
    print(f'M={M}  one-kernel fwd {t_f:.1f} us ({nb_f / t_f / 1e6:.2f} TB/s)  bwd {t_b:.1f} us ({nb_b / t_b / 1e6:.2f} TB/s)')
    print(f'       round-5 fwd {t_f1:.1f} + {t_f2:.1f} = {t_f1 + t_f2:.1f} us   bwd {t_b1:.1f} + {t_b2:.1f} + {t_b3:.1f} = {t_b1 + t_b2 + t_b3:.1f} us')


if __name__ == '__main__':
    main()
