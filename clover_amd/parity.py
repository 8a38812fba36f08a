"""Parity mode: the Clover step with fp32 storage and fp32 arithmetic on the HIP kernels.

north_star asks for step losses within 1e-3 of the reference's CPU path, which is fp32
(mmaction/models/recognizers/multimodal_transformer_pretrain.py:129-169; contrastive_loss.py:102-161 divides
cosines by the temperature 0.05, a x20 amplification).  The training path computes on bf16 MFMA operands and stores
bf16 activations; that carries 2^-9 relative rounding per stage and lands at 1e-3 .. 1e-2 on the contrastive losses.
Parity mode runs the SAME registered modules, window geometry, token maps, bias-table indexing, masks, heads and loss
kernels, but
  * every activation is stored in fp32,
  * every Linear is ``clv_sgemm_nt`` (exact-f32 MFMA 16x16x4) on the fp32 master weights,
  * attention is ``clv_attn_f32_fwd`` (fp32 VALU arithmetic through the index helpers the MFMA kernels use),
  * LayerNorm / GELU are the production kernels in their fp32-storage instantiation,
so that what is compared with the reference at 1e-3 is the path's structure and index logic with the bf16 rounding
taken out.  Forward AND backward: every op above has an fp32 backward on hand-written kernels (clv_sgemm_nt on transposed
operands, clv_attn_f32_bwd, the LayerNorm / GELU backward kernels in fp32 storage), so the step's gradients are checked
against the reference's at fp32 tolerances too (tests/test_parity_gpu.py).

Switch: ``CLOVER_PARITY=1`` in the environment or ``with parity.mode():``.  ``round=`` re-injects single bf16 rounding
sources (error isolation): 'act' (activations written by a kernel), 'stream' (the Swin residual stream),
'weight' (GEMM weights), 'prob' (attention probabilities), 'input' (the clip as the patch-embedding GEMM operand).
"""
import contextlib
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ClvAttnGeom, check

BF16 = torch.bfloat16
ROUND_KINDS = ('act', 'stream', 'weight', 'prob', 'input')


def _env_state():
    if os.environ.get('CLOVER_PARITY', '0') in ('', '0'):
        return None
    kinds = [k for k in os.environ.get('CLOVER_PARITY_ROUND', '').split(',') if k]
    assert all(k in ROUND_KINDS for k in kinds), kinds
    return dict(round=frozenset(kinds), dtype=BF16)


STATE = _env_state()


def enabled():
    return STATE is not None


@contextlib.contextmanager
def mode(round=(), dtype=BF16):
    """Run the enclosed forward passes in parity mode (see module docstring).  dtype: what the re-injected rounding sources
    round TO — bf16 (the training path's storage / operand type) or torch.float16 (round 6: what the step's losses would
    look like with f16 storage and f16 MFMA operands, the reference's own arithmetic type, without building those kernels)."""
    global STATE
    assert all(k in ROUND_KINDS for k in round), round
    assert dtype in (BF16, torch.float16)
    prev, STATE = STATE, dict(round=frozenset(round), dtype=dtype)
    try:
        yield
    finally:
        STATE = prev


def rnd(kind, t):
    """Re-inject one bf16 rounding source (isolation runs); identity in plain parity mode."""
    if t is None or kind not in STATE['round']:
        return t
    return t.to(STATE.get('dtype', BF16)).float()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def f32(t):
    """fp32 storage of an activation (what to_bf16 is on the training path)."""
    return rnd('act', t.float())


def sgemm(a, b, bias=None):
    """c [M,N] fp32 = a [M,K] . b [N,K]^T + bias (clv_sgemm_nt)."""
    assert a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32
    a = a if a.stride(-1) == 1 else a.contiguous()
    b = b if b.stride(-1) == 1 else b.contiguous()
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    c = torch.empty(M, N, device=a.device, dtype=torch.float32)
    bf = bias.detach().float().contiguous() if bias is not None else None
    check(_lib.lib().clv_sgemm_nt(_ptr(a), _ptr(b), _ptr(bf), _ptr(c), M, N, K, a.stride(0), b.stride(0), N, _stream()),
          'clv_sgemm_nt')
    return c


class _LinearP(torch.autograd.Function):
    """y = x W^T + b on clv_sgemm_nt, forward and backward (dx = dy W, dW = dy^T x, db = sum dy: the same exact-f32 MFMA
    kernel on transposed copies — parity mode is about values, not speed).  Gradients go back to autograd in fp32."""

    @staticmethod
    def forward(ctx, x2, weight, bias):
        w = rnd('weight', weight.detach().float())
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        return sgemm(x2, w, bias)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy = dy.float().contiguous()
        dx = sgemm(dy, w.t().contiguous()) if ctx.needs_input_grad[0] else None           # [M,N] . ([K,N])^T
        dw = sgemm(dy.t().contiguous(), x2.t().contiguous()) if ctx.needs_input_grad[1] else None   # [N,M] . ([K,M])^T
        db = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def linear(x, weight, bias):
    K = weight.shape[1]
    x2 = x.float().reshape(-1, K)
    y = _LinearP.apply(x2, weight, bias)
    return rnd('act', y).view(x.shape[:-1] + (weight.shape[0],))


class _AttentionP(torch.autograd.Function):
    """clv_attn_f32_fwd / clv_attn_f32_bwd on a packed fp32 q|k|v tensor; the relative-position table's gradient is returned
    to autograd (fp32, the table's shape)."""

    @staticmethod
    def forward(ctx, qkv, table, rid, kmask, geom_kw, round_p):
        g = ClvAttnGeom(**geom_kw)
        Cdim = g.nH * g.hd
        g.ldq = g.ldk = g.ldv = 3 * Cdim
        g.ldo = Cdim
        tab = table.detach().float().contiguous() if table is not None else None
        km = kmask.float().contiguous() if kmask is not None else None
        o = torch.empty(qkv.shape[:-1] + (Cdim,), device=qkv.device, dtype=torch.float32)
        base = qkv.data_ptr()
        check(_lib.lib().clv_attn_f32_fwd(C.c_void_p(base), C.c_void_p(base + 4 * Cdim), C.c_void_p(base + 8 * Cdim), _ptr(o),
                                          _ptr(tab), _ptr(rid), _ptr(km), C.byref(g), int(round_p), _stream()),
              'clv_attn_f32_fwd')
        ctx.save_for_backward(qkv, tab, rid, km)
        ctx.geom, ctx.has_table = g, table is not None
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, tab, rid, km = ctx.saved_tensors
        g = ctx.geom
        Cdim = g.nH * g.hd
        do = do.float().contiguous()
        dqkv = torch.empty_like(qkv)
        dtab = torch.zeros_like(tab) if tab is not None else None
        L = _lib.lib()
        work = torch.empty(L.clv_attn_f32_bwd_work_floats(C.byref(g)), device=qkv.device, dtype=torch.float32)
        b, d = qkv.data_ptr(), dqkv.data_ptr()
        check(L.clv_attn_f32_bwd(C.c_void_p(b), C.c_void_p(b + 4 * Cdim), C.c_void_p(b + 8 * Cdim), _ptr(do), _ptr(tab),
                                 _ptr(rid), _ptr(km), C.c_void_p(d), C.c_void_p(d + 4 * Cdim), C.c_void_p(d + 8 * Cdim),
                                 _ptr(dtab), _ptr(work), C.byref(g), _stream()), 'clv_attn_f32_bwd')
        return dqkv, (dtab if ctx.has_table else None), None, None, None, None


def attention(qkv, table, rid, kmask, geom_kw):
    """clv_attn_f32_fwd (+ its fp32 backward) on a packed fp32 [..., 3*nH*hd] q|k|v tensor -> o fp32 [..., nH*hd]."""
    assert qkv.is_cuda
    qkv = qkv.float().contiguous()
    if geom_kw.get('dropout_p'):
        raise NotImplementedError('parity mode is an eval-mode pass (no attention dropout)')
    o = _AttentionP.apply(qkv, table, rid, kmask, geom_kw,
                          (2 if STATE.get('dtype', BF16) == torch.float16 else 1) if 'prob' in STATE['round'] else 0)
    return rnd('act', o)


def patch_embed(x, weight, bias, gamma, beta, mask_token, vmask, want_clean, eps, stacked=False):
    """PatchEmbed3D (swin_transformer_3d.py:665-688: Conv3d(kernel = stride = (2,4,4)) + LayerNorm) and the mask-token
    blend (:226-229) in fp32: im2col view -> clv_sgemm_nt -> fp32 LayerNorm kernel -> blend.
    Returns (clean, masked) fp32 channels-last [B,T/2,H/4,W/4,C] (or the stacked [2B,...] tensor)."""
    from . import ops
    B, Cin, T, H, W = x.shape
    Cout = weight.shape[0]
    assert Cin == 3 and tuple(weight.shape[1:]) == (3, 2, 4, 4)
    if T % 2 or H % 4 or W % 4:                            # right / bottom / back zero-pad to patch multiples (:679-680);
        x = torch.nn.functional.pad(x, (0, -W % 4, 0, -H % 4, 0, -T % 2))      # PatchEmbed3D.pad has normally done it
        B, Cin, T, H, W = x.shape
    Tp, Hp, Wp = T // 2, H // 4, W // 4
    # patches [M, 96] in the weight's (c, dt, dh, dw) column order: a pure layout op
    patches = (rnd('input', x.float()).reshape(B, 3, Tp, 2, Hp, 4, Wp, 4).permute(0, 2, 4, 6, 1, 3, 5, 7)
               .reshape(B * Tp * Hp * Wp, 96))
    z = _LinearP.apply(patches, weight.reshape(Cout, 96), bias)          # rounds the weight itself when asked to
    if gamma is not None:
        z = ops.layer_norm(z, gamma, beta, eps)            # parity branch of ops.layer_norm: fp32 kernel
    clean = rnd('act', z).view(B, Tp, Hp, Wp, Cout)
    masked = None
    if vmask is not None:
        mh, mw = vmask.shape[-2], vmask.shape[-1]
        w = vmask.reshape(B, 1, mh, 1, mw, 1).expand(B, Tp, mh, Hp // mh, mw, Wp // mw).reshape(B, Tp, Hp, Wp, 1)
        w = w.to(torch.float32)
        masked = rnd('act', z.view(B, Tp, Hp, Wp, Cout) * (1.0 - w) + mask_token.float().reshape(1, 1, 1, 1, Cout) * w)
    if stacked:
        return torch.cat([clean, masked], 0)
    return (clean if want_clean else None), masked
