"""fp8 forward-GEMM path (BASELINE config 5; clv_quant_fp8_rows + clv_gemm_nt_fp8 under CLOVER_FP8=1): the kernels against
fp32 restatements with stated tolerances, autograd through the fp8 Linear / MLP (bf16 gradients), and the step losses
against the oracle.  `-m gpu` only."""
import os
import sys

import pytest
import torch

from clover_amd import _lib as _clv_lib  # noqa: E402

# fp8 forward GEMMs quantise bf16 tensors (BASELINE config 5: "fp8 MFMA QKV / patch-proj path" over the bf16 step): this file
# runs in the bf16 build of the kernels — in the default (fp16) test process it is skipped and tests/test_bf16_build_gpu.py
# runs it in a child process with CLOVER_HALF=bf16
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(_clv_lib.HALF_F16, reason='fp8 path: bf16 build (see test_bf16_build_gpu.py)')]
DEV = 'cuda'
BF = torch.bfloat16
F8 = torch.float8_e4m3fn


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize('M,K', [(777, 768), (64, 128), (5, 3072), (1000, 1000), (33, 4096)])
def test_quant_fp8_rows(M, K):
    """q = e4m3(x / s), s = rowmax|x| / 448: scales exact, bytes equal to torch's OCP e4m3 cast of the same quotient
    (an all-zero row keeps s = 1)."""
    from clover_amd import ops
    x = rnd(M, K, seed=1).to(BF)
    x[M // 2] = 0
    q, s = ops.quant_fp8_rows(x.to(DEV))
    amax = x.float().abs().amax(1)
    s_ref = torch.where(amax > 0, amax / 448.0, torch.ones_like(amax))
    assert torch.allclose(s.cpu(), s_ref, rtol=1e-6, atol=0)
    inv = torch.where(amax > 0, 448.0 / amax, torch.ones_like(amax))
    q_ref = (x.float() * inv[:, None]).clamp(-448, 448).to(F8)
    deq, deq_ref = q.cpu().view(F8).float(), q_ref.float()
    assert (deq != deq_ref).float().mean().item() < 5e-3          # x * inv next to a rounding tie may land on the other side
    assert rel(deq * s.cpu()[:, None], x.float()) < 2 ** -4 + 1e-3      # 3 mantissa bits: half an ulp = 2^-4 relative


@pytest.mark.parametrize('M,N,K', [(3136, 2304, 768), (512, 768, 3072), (12544, 384, 1536), (777, 136, 256), (64, 64, 128),
                                   (25088, 1024, 256)])
@pytest.mark.parametrize('epi', ['none', 'bias', 'gelud'])
def test_gemm_nt_fp8(M, N, K, epi):
    """clv_gemm_nt_fp8 against (i) the fp32 product of the DEQUANTISED operands — the kernel's own arithmetic: bf16 output
    rounding only, 1e-2 of max — and (ii) the fp32 product of the original operands — what the e4m3 operands cost: 5e-2 of
    max (3 mantissa bits per element, averaged over the contraction)."""
    from clover_amd import ops
    a, b = rnd(M, K, seed=3).to(BF), (rnd(N, K, seed=4) * 0.05).to(BF)
    bias = rnd(N, seed=5) if epi != 'none' else None
    ad, bd = a.to(DEV), b.to(DEV)
    aq, asc = ops.quant_fp8_rows(ad)
    bq, bsc = ops.quant_fp8_rows(bd)
    deq = lambda q, s: q.view(F8).float() * s[:, None]
    pre_q = deq(aq, asc) @ deq(bq, bsc).t() + (bias.to(DEV) if bias is not None else 0)
    pre_x = ad.float() @ bd.float().t() + (bias.to(DEV) if bias is not None else 0)
    e = {'none': ops.GEMM_EPI_NONE, 'bias': ops.GEMM_EPI_BIAS, 'gelud': ops.GEMM_EPI_BIAS_GELU_D}[epi]
    out = ops.gemm_nt_fp8(ad, bd, bias.to(DEV) if bias is not None else None, epilogue=e)
    if epi == 'gelud':
        c, d = out
        xr = pre_q.clone().requires_grad_()
        yr = torch.nn.functional.gelu(xr)
        yr.sum().backward()
        assert rel(c, yr) < 1e-2 and rel(d, xr.grad) < 1e-2, (rel(c, yr), rel(d, xr.grad))
        assert rel(c, torch.nn.functional.gelu(pre_x)) < 5e-2
    else:
        assert rel(out, pre_q) < 1e-2, rel(out, pre_q)
        assert rel(out, pre_x) < 5e-2, rel(out, pre_x)


def test_fp8_linear_and_mlp_autograd(monkeypatch):
    """ops.linear / ops.mlp_gelu with the fp8 forward: outputs within the fp8 tolerance of fp32, gradients (bf16 backward on
    the bf16 activations / weights) within the bf16 tolerance of the fp32 gradients taken at the fp32 forward."""
    from clover_amd import ops
    monkeypatch.setattr(ops, 'FP8', True)
    M, C, Hd = 8192, 256, 1024
    monkeypatch.setenv('CLOVER_FP8_MIN_N', '64')
    x = rnd(M, C, seed=7).to(BF)
    w1, b1 = rnd(Hd, C, seed=8) * 0.05, rnd(Hd, seed=9) * 0.1
    w2, b2 = rnd(C, Hd, seed=10) * 0.05, rnd(C, seed=11) * 0.1
    dy = rnd(M, C, seed=12).to(BF)
    xr = x.float().requires_grad_()
    pr = [t.clone().requires_grad_() for t in (w1, b1, w2, b2)]
    yr = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xr, pr[0], pr[1])), pr[2], pr[3])
    yr.backward(dy.float())
    xg = x.to(DEV).requires_grad_()
    pg = [t.clone().to(DEV).requires_grad_() for t in (w1, b1, w2, b2)]
    assert ops.fp8_ok(xg.detach(), Hd, C) and ops.mlp_gelu_ok(xg.detach(), Hd)
    y = ops.mlp_gelu(xg, *pg)
    y.backward(dy.to(DEV))
    assert rel(y, yr) < 6e-2, rel(y, yr)
    assert rel(xg.grad, xr.grad) < 8e-2, rel(xg.grad, xr.grad)
    for g, r in zip(pg, pr):
        assert rel(g.grad, r.grad) < 8e-2, rel(g.grad, r.grad)
    xl = x.to(DEV).requires_grad_()
    wl, bl = w1.clone().to(DEV).requires_grad_(), b1.clone().to(DEV).requires_grad_()
    z = ops.linear(xl, wl, bl)
    z.backward(torch.ones_like(z))
    zr = torch.nn.functional.linear(x.float(), w1, b1)
    assert rel(z, zr) < 5e-2
    assert rel(wl.grad, torch.ones(M, Hd).t() @ x.float()) < 2e-2         # the backward is the bf16 one


@pytest.mark.parametrize('rows,C', [(3136, 768), (777, 384), (100, 1024), (64, 3072), (50, 256)])
def test_layernorm_emits_fp8_operand(rows, C, monkeypatch):
    """With the fp8 path on, the LayerNorm forward also writes its output as e4m3 + row scales (the next GEMM's operand):
    scale = rowmax|y| / 448 and the de-quantised rows equal y to half an e4m3 ulp; the consumer takes the pair instead of
    running clv_quant_fp8_rows (same GEMM result as with the separate pass, up to the bf16 rounding of y the pass sees)."""
    from clover_amd import ops
    monkeypatch.setattr(ops, 'FP8', True)
    x = rnd(rows, C, seed=31).to(BF).to(DEV)
    r = rnd(rows, C, seed=32).to(BF).to(DEV)
    g, b = (1 + 0.1 * rnd(C, seed=33)).to(DEV), (0.1 * rnd(C, seed=34)).to(DEV)
    y, ssum = ops.layer_norm(x, g, b, 1e-5, residual=r, return_sum=True)
    assert hasattr(y, '_clv_fp8')
    q, sc = ops._fp8_side(y)
    assert q.shape == (rows, C) and q.dtype == torch.uint8
    amax = y.float().abs().amax(1)
    assert torch.allclose(sc, amax / 448.0, rtol=2 ** -7)                       # y is the bf16 rounding of what was scaled
    deq = q.view(F8).float() * sc[:, None]
    assert rel(deq, y.float()) < 2 ** -4 + 2 ** -7
    w = (rnd(1536, C, seed=35) * 0.05).to(BF).to(DEV)
    c_fused = ops.gemm_nt_fp8(y, w, None, aq8=(q, sc))
    c_pass = ops.gemm_nt_fp8(y, w, None)
    assert rel(c_fused, c_pass) < 4e-2            # fp32 y vs its bf16 rounding as the quantised value: e4m3 ties flip
    # the pair is bound to y's version: an in-place write between the LayerNorm and its Linear drops it
    yw = ops.layer_norm(x, g, b, 1e-5, residual=r)
    assert ops._fp8_side(yw) is not None
    yw.mul_(2.0)
    assert ops._fp8_side(yw) is None
    monkeypatch.setattr(ops, 'FP8', False)
    y2 = ops.layer_norm(x, g, b, 1e-5, residual=r)
    assert not hasattr(y2, '_clv_fp8') and torch.equal(y2, y)


@pytest.mark.parametrize('C,B,T,HW', [(96, 2, 8, 56), (128, 1, 4, 56), (48, 2, 4, 112)])
def test_patch_embed_fp8(C, B, T, HW, monkeypatch):
    """The patch projection on the fp8 matrix instruction (per-token and per-output-channel scales, in-kernel e4m3
    conversion of the clip) + bias + LayerNorm + blend against the fp32 module: 6e-2 of max after the LayerNorm (the
    bf16 kernel holds 2e-2 on the same check), masked / clean consistency exact."""
    import torch.nn.functional as F
    from clover_amd import ops
    x = rnd(B, 3, T, HW, HW, seed=61)
    w, b = rnd(C, 3, 2, 4, 4, seed=62) * 0.1, rnd(C, seed=63) * 0.1
    g, be = 1 + 0.1 * rnd(C, seed=64), 0.1 * rnd(C, seed=65)
    mt = rnd(1, C, 1, 1, 1, seed=66) * 0.02
    vm = (torch.rand(B, 1, 7, 7, generator=torch.Generator().manual_seed(67)) < 0.3).long()
    z = F.conv3d(x, w, b, stride=(2, 4, 4)).permute(0, 2, 3, 4, 1)
    ref = F.layer_norm(z, (C,), g, be, 1e-5)
    monkeypatch.setattr(ops, 'FP8', True)
    clean, masked = ops.patch_embed(x.to(DEV), w.to(DEV), b.to(DEV), g.to(DEV), be.to(DEV), mt.to(DEV), vm.to(DEV))
    assert rel(clean, ref) < 6e-2, rel(clean, ref)
    wmap = vm.reshape(B, 1, 7, 1, 7, 1).expand(B, T // 2, 7, HW // 28, 7, HW // 28).reshape(B, T // 2, HW // 4, HW // 4, 1).bool()
    sel = wmap.expand_as(ref).to(DEV)
    assert torch.equal(masked[~sel], clean[~sel])                                  # untouched tokens identical
    assert rel(masked[sel].view(-1, C), mt.view(1, C).expand(int(wmap.sum()), C)) < 1e-2


# |loss - oracle| of the fp8-forward step, B = 2, eval mode: ~2x the largest value measured over the three
# configurations below (printed by the test; round 4 numbers in DESIGN.md section 5)
FP8_LOSS_TOL = dict(mlm_loss=1e-2, nce_loss=5e-2, rank_t_tm_loss=1.2e-1, v_nce_loss=2.7e-1, rank_v_vm_loss=1.7e-1, loss=4.5e-1)
# max|grad - oracle| / max|oracle| of gradients that cross every encoder (fp8 forward operands, bf16 backward)
FP8_GRAD_TOL = 0.30         # measured <= 17.5 % (e4m3 operands carry 2^-4 relative rounding)


@pytest.mark.usefixtures('strict_own_gemm')
@pytest.mark.parametrize('variant,frames', [('T', 8), ('B', 16), ('B', 32)])
def test_fp8_step_losses_vs_oracle(variant, frames, monkeypatch):
    """The step with fp8 forward GEMMs (Swin stages 0-3, text tower, fusion encoder) against the fp32 oracle, B = 2, eval
    mode — BASELINE config 2's shapes, config 4's (Swin-B, 16 frames) and CONFIG 5's (Swin-B, 32 frames: (8,7,7) windows
    with the (4,3,3) shift, 816-token fusion sequences).  Losses within FP8_LOSS_TOL, and the gradients of
    test_step_gpu.FULL_GRAD_KEYS (7 / 10 parameters across the video encoder, the text tower, the fusion encoder and the
    MLM head) within FP8_GRAD_TOL of the oracle's, relative to the gradient's max magnitude."""
    import gutil
    import clover_amd
    from clover_amd import ops
    from test_step_gpu import FULL_GRAD_KEYS
    cfg, sd, batch, lv_ref, gref = gutil.full_size_oracle(variant, frames)
    m = clover_amd.build_model(cfg).eval()
    m.load_state_dict(sd)
    m = m.to(DEV)
    monkeypatch.setattr(ops, 'FP8', True)
    out = m.train_step({k: v.to(DEV) for k, v in batch.items()}, None)
    lv = out['log_vars']
    errs = {k: abs(lv[k] - lv_ref[k]) for k in FP8_LOSS_TOL}
    print(f'fp8 loss errors Swin-{variant} {frames}f', errs)
    for k, tol in FP8_LOSS_TOL.items():
        assert errs[k] <= tol, (k, lv[k], lv_ref[k])
    out['loss'].backward()
    named = dict(m.named_parameters())
    worst = {k: rel(named[k].grad, gref[k]) for k in FULL_GRAD_KEYS[variant]}
    print(f'fp8 grad rel errors Swin-{variant} {frames}f', worst)
    for k, e in worst.items():
        assert e < FP8_GRAD_TOL, (k, e)
