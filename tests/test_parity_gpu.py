"""Parity mode (clover_amd/parity.py): the step with fp32 storage + fp32 arithmetic on the HIP kernels, asserted at
the north-star tolerance — all six step losses within 1e-3 of the REFERENCE's own numbers (goldens, config 1) and of the
oracle at BASELINE config 2's and config 4's shapes — plus the fp32 kernels against fp64 torch restatements built from
the oracle's index helpers.  `-m gpu` only."""
import os
import sys

import numpy as np
import pytest
import torch

import closed_form as cf
import gutil

pytestmark = pytest.mark.gpu

from oracle import indexing as ix          # noqa: E402
from oracle import model as om             # noqa: E402

DEV = 'cuda'
LOSS_KEYS = ['mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss']
PARITY_TOL = 1e-3                      # north_star: "losses/logits within 1e-3"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ----------------------------------------------------------------------------- kernels
@pytest.mark.parametrize('M,N,K', [(777, 288, 96), (64, 64, 16), (1000, 30522, 768), (50, 100, 52), (3, 5, 7),
                                   (6272, 1152, 384), (1, 768, 3072)])
@pytest.mark.parametrize('bias', [True, False])
def test_sgemm_nt(M, N, K, bias):
    from clover_amd import parity
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bb = rnd(N, seed=3) if bias else None
    ref = a.double() @ b.double().t() + (bb.double() if bias else 0)
    with parity.mode():
        c = parity.sgemm(a.to(DEV), b.to(DEV), bb.to(DEV) if bias else None)
    assert c.dtype == torch.float32 and rel(c, ref) < 1e-5, rel(c, ref)      # an fp32 fmaf chain over K <= 3072


def test_sgemm_nt_strided_rows():
    from clover_amd import parity
    a = rnd(100, 200, seed=4).to(DEV)[:, :96]               # lda = 200
    b = rnd(60, 96, seed=5).to(DEV)
    with parity.mode():
        c = parity.sgemm(a, b)
    assert rel(c, a.double() @ b.double().t()) < 2e-6


@pytest.mark.parametrize('M,N,K,bias', [(16, 1536, 768, True), (8, 768, 1536, True), (3, 130, 70, False), (40, 768, 768, True),
                                          (3, 130, 70, True), (100, 96, 64, True)])     # (> 96 rows: bias gradient as its own launch)
def test_linear_f32_heads(M, N, K, bias):
    """ops.linear_f32 (clv_sgemm_strided: the fp32 Linear of the contrastive projection heads, on the TRAINING path):
    forward, input gradient, weight / bias gradient against fp64 torch — returned to autograd and accumulated into
    engine-style fp32 sinks."""
    from clover_amd import ops
    x, w = rnd(M, K, seed=51), rnd(N, K, seed=52) * 0.05
    b = rnd(N, seed=53) if bias else None
    dy = rnd(M, N, seed=54)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    br = b.double().requires_grad_() if bias else None
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(dy.double())
    xg, wg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
    bg = b.to(DEV).requires_grad_() if bias else None
    y = ops.linear_f32(xg, wg, bg)
    y.backward(dy.to(DEV))
    assert y.dtype == torch.float32 and rel(y, yr) < 1e-5
    assert rel(xg.grad, xr.grad) < 1e-5 and rel(wg.grad, wr.grad) < 1e-5
    if bias:
        assert rel(bg.grad, br.grad) < 1e-5
    # sinks: the gradients are ADDED into pre-filled fp32 buffers and nothing is returned to autograd
    w2, b2 = w.to(DEV).requires_grad_(), (b.to(DEV).requires_grad_() if bias else None)
    w0, b0 = rnd(N, K, seed=55).to(DEV), rnd(N, seed=56).to(DEV)
    w2._clv_grad, w2._clv_ready = w0.clone(), (lambda: None)
    if bias:
        b2._clv_grad, b2._clv_ready = b0.clone(), (lambda: None)
    ops.linear_f32(x.to(DEV).requires_grad_(), w2, b2).backward(dy.to(DEV))
    assert w2.grad is None and rel(w2._clv_grad, w0.double().cpu() + wr.grad) < 1e-5
    if bias:
        assert rel(b2._clv_grad, b0.double().cpu() + br.grad) < 1e-5


WIN_CASES = [
    (2, 4, 14, 14, 96, 3, False), (2, 4, 14, 14, 96, 3, True), (1, 2, 14, 14, 48, 3, True), (2, 4, 7, 7, 64, 2, True),
    (1, 16, 14, 14, 64, 2, True), (1, 4, 14, 14, 128, 2, True),
]


@pytest.mark.parametrize('case', WIN_CASES)
def test_window_attention_f32(case):
    """clv_attn_f32_fwd (mode 1) == roll + partition + WindowAttention3D + reverse + un-roll (swin_transformer_3d.py:375-397,
    459-476) in fp64, over the oracle's index helpers: token map, relative_position_index[:N,:N], compute_mask."""
    from test_kernels_gpu import ref_window_attention
    from clover_amd import ops, parity
    from clover_amd.backbones.swin_transformer_3d import window_geometry
    B, D, H, W, C, nH, shifted = case
    cfg_ws, cfg_ss = (8, 7, 7), ((4, 3, 3) if shifted else (0, 0, 0))
    qkv = rnd(B, D, H, W, 3 * C, seed=11)
    table = rnd((2 * 8 - 1) * 13 * 13, nH, scale=0.5, seed=12)
    o_ref = ref_window_attention(qkv.double(), table.double(), ix.relative_position_index(cfg_ws), cfg_ws, cfg_ss, nH)
    ws, ss, rid = window_geometry((D, H, W), cfg_ws, cfg_ss, DEV)
    with parity.mode():
        o = ops.window_attention(qkv.to(DEV), table.to(DEV), rid, ws, ss, nH, table_window=cfg_ws)
    assert o.dtype == torch.float32 and rel(o, o_ref) < 1e-5, rel(o, o_ref)


@pytest.mark.parametrize('B,S,nH,hd', [(3, 16, 2, 64), (2, 228, 12, 64), (2, 40, 4, 32), (2, 816, 12, 64), (1, 1000, 2, 16)])
def test_seq_attention_f32(B, S, nH, hd):
    from clover_amd import ops, parity
    Hd = nH * hd
    qkv = rnd(B, S, 3 * Hd, seed=21)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[0, S - 5:] = 0
    ext = om.extended_mask(mask)
    q, k, v = qkv.double().view(B, S, 3, nH, hd).permute(2, 0, 3, 1, 4)
    p = (q @ k.transpose(-1, -2) / hd ** 0.5 + ext.double()).softmax(-1)
    o_ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    with parity.mode():
        o = ops.seq_attention(qkv.to(DEV), ext.reshape(B, S).to(DEV).contiguous(), nH)
    assert o.dtype == torch.float32 and rel(o, o_ref) < 1e-5, rel(o, o_ref)


# ----------------------------------------------------------------------------- the step, config 1: reference goldens
@pytest.fixture(scope='module')
def model():
    import clover_amd
    m = clover_amd.build_model(cf.tiny_model_cfg())
    m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False)
    return m.to(DEV).eval()


@pytest.mark.parametrize('B', [1, 2, 4])
def test_parity_step_losses_vs_reference(model, B):
    """BASELINE config 1, closed-form weights: the six losses of train_step in parity mode against the numbers the
    REFERENCE itself produced (tests/golden/g_step.npz), at 1e-3."""
    from clover_amd import parity
    g = gutil.load('g_step.npz')
    batch = {k: v.to(DEV) for k, v in cf.cf_batch(B, tag=f'step{B}').items()}
    with parity.mode(), torch.no_grad():
        lv = model.train_step(batch, None)['log_vars']
    errs = {k: abs(lv[k] - float(g[f'B{B}.{k}'])) for k in LOSS_KEYS}
    print('parity loss errors (config 1)', B, errs)
    for k in LOSS_KEYS:
        assert errs[k] <= PARITY_TOL, (k, lv[k], float(g[f'B{B}.{k}']))


def test_parity_isolation_per_rounding_source(model):
    """tools/parity_isolate.py as a test (ADVICE r3 / VERDICT r4 item 7): parity mode with ONE bf16 rounding source re-injected
    at a time (activations a kernel writes / the Swin residual stream / GEMM weights / attention probabilities / the clip as
    patch-embedding operand), config 1, B = 2 and 4, against the reference's own losses.  Claims pinned: (i) plain parity
    mode holds 1e-3; (ii) every single source stays inside the bf16 path's tolerances (tests/test_step_gpu.py LOSS_TOL);
    (iii) bf16 activation storage or bf16 weights ALONE already move a contrastive / rank loss past 1e-3 — which is why the
    shipped bf16 path cannot meet the north-star bound and the fp32 parity mode exists; (iv) all five together land within
    the bf16 tolerances of the reference, i.e. the emulation accounts for the training path's error."""
    from clover_amd import parity
    from test_step_gpu import LOSS_TOL_BF16 as LOSS_TOL          # the re-injected roundings are bf16 ones
    g = gutil.load('g_step.npz')
    broke = set()
    for B in (2, 4):
        batch = {k: v.to(DEV) for k, v in cf.cf_batch(B, tag=f'step{B}').items()}
        ref = {k: float(g[f'B{B}.{k}']) for k in LOSS_KEYS}

        def errs(rounds):
            with parity.mode(round=rounds), torch.no_grad():
                lv = model.train_step(batch, None)['log_vars']
            return {k: abs(lv[k] - ref[k]) for k in LOSS_KEYS}
        e0 = errs(())
        assert max(e0.values()) <= PARITY_TOL, e0
        for kind in parity.ROUND_KINDS:
            e = errs((kind,))
            print('parity +', kind, B, {k: f'{v:.1e}' for k, v in e.items()})
            for k in LOSS_KEYS:
                assert e[k] <= LOSS_TOL[k], (kind, k, e[k])
            if max(e[k] for k in LOSS_KEYS if k != 'mlm_loss') > PARITY_TOL:
                broke.add(kind)
        ea = errs(parity.ROUND_KINDS)
        for k in LOSS_KEYS:
            assert ea[k] <= LOSS_TOL[k], ('all', k, ea[k])
    assert {'act', 'weight'} & broke, broke


def test_parity_modules_vs_reference(model):
    """Feature maps of the three encoders in parity mode against the reference goldens: 1e-4 of max (the bf16 path
    asserts 2-3e-2 on the same fixtures)."""
    from clover_amd import parity
    g = gutil.load('g_swin.npz')
    b = cf.cf_batch(2, tag='swin')
    x, vm = b['imgs'][:, 0].to(DEV), b['v_token_mask'].to(DEV)
    with parity.mode(), torch.no_grad():
        y = model.backbone(x)
        ym, w = model.backbone(x.clone(), vm)
    assert rel(y, torch.from_numpy(g['clean.out'])) < 1e-4
    assert rel(ym, torch.from_numpy(g['masked.out'])) < 1e-4
    assert np.array_equal(w.float().cpu().numpy().astype(np.int8), g['masked.w'])
    g = gutil.load('g_bert_fuse.npz')
    b = cf.cf_batch(3, tag='bf')
    ids, mask = b['token_ids'][:, 0].to(DEV), b['input_mask'][:, 0].to(DEV)
    with parity.mode(), torch.no_grad():
        t = model.text_backbone(ids, mask)['last_hidden_state']
        vt = cf.cf_float('bf.vt', (3, 2, 196, 96), 1.0).to(DEV)
        tt = torch.from_numpy(g['bert.last_hidden_state']).to(DEV)
        f = model.multimodal_backbone(visual_token=vt, text_input_mask=mask, text_input_embeds=tt)
    assert rel(t, torch.from_numpy(g['bert.last_hidden_state'])) < 1e-4
    assert rel(f['t_last_hidden_state'], torch.from_numpy(g['fuse.t_last_hidden_state'])) < 1e-4


# ----------------------------------------------------------------------------- full-size shapes: oracle
def _full_size(variant, frames, B, seed):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import clover_amd
    torch.manual_seed(seed)
    cfg = bench.model_cfg(variant, frames)
    m = clover_amd.build_model(cfg).eval()
    P = {k: v.detach().float() for k, v in m.state_dict().items() if 'relative_position_index' not in k}
    batch = bench.synthetic_batch(B, frames, 32, seed=seed + 1)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        _, lv_ref = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
    return m.to(DEV), {k: v.to(DEV) for k, v in batch.items()}, lv_ref


@pytest.mark.parametrize('variant,frames,B', [('T', 8, 2), ('B', 16, 2), ('T', 32, 2)])
def test_parity_full_size_losses_vs_oracle(variant, frames, B):
    """BASELINE config 2's (Swin-T, 8 frames), config 4's (Swin-B, 16 frames: 392-token windows, fc_in 1024 -> 768,
    the depth-18 stage) and config 5's clip length (32 frames: temporal shift, 816-token fusion sequences) shapes,
    seeded init: the six losses in parity mode against the fp32 oracle at 1e-3."""
    from clover_amd import parity
    m, batch, lv_ref = _full_size(variant, frames, B, 4321 + frames)
    with parity.mode(), torch.no_grad():
        lv = m.train_step(batch, None)['log_vars']
    errs = {k: abs(lv[k] - lv_ref[k]) for k in LOSS_KEYS}
    print(f'parity loss errors Swin-{variant} {frames}f B={B}', errs, {k: lv_ref[k] for k in LOSS_KEYS})
    for k in LOSS_KEYS:
        assert errs[k] <= PARITY_TOL, (k, lv[k], lv_ref[k])


def test_parity_one_frame_image_batch(model):
    """1-frame image batches (CC3M samples, SURVEY Appendix A: PatchEmbed3D zero-pads the depth 1 -> 2,
    swin_transformer_3d.py:679-680) run in parity mode too — the patch embedding pads like the module does — and meet the
    1e-3 bound against the oracle."""
    from clover_amd import parity
    batch = cf.cf_batch(2, frames=1, tag='par_img')
    P = cf.cf_state(gutil.manifest())
    with torch.no_grad():
        _, lv_ref = om.parse_losses(om.forward_train(P, batch, cf.oracle_cfg_from(cf.tiny_model_cfg()), gather=False))
        with parity.mode():
            lv = model.train_step({k: v.to(DEV) for k, v in batch.items()}, None)['log_vars']
            # the raw (un-padded) clip straight into the parity patch embedding: it pads by itself
            x = batch['imgs'][:, 0].to(DEV)
            pe = model.backbone.patch_embed
            a, _ = parity.patch_embed(x, pe.proj.weight, pe.proj.bias, pe.norm.weight, pe.norm.bias, None, None, True,
                                      pe.norm.eps)
            b, _ = pe.tokens(x)
    errs = {k: abs(lv[k] - lv_ref[k]) for k in LOSS_KEYS}
    print('parity 1-frame loss errors', errs)
    for k in LOSS_KEYS:
        assert errs[k] <= PARITY_TOL, (k, lv[k], lv_ref[k])
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize('B', [1, 2, 4])
def test_parity_step_gradients_vs_reference(model, B):
    """Parity mode has a backward (fp32 storage + arithmetic on the HIP kernels: clv_sgemm_nt on transposed operands,
    clv_attn_f32_bwd, the LayerNorm / GELU backward kernels in fp32 storage): the 13 parameter gradients the REFERENCE
    itself produced for BASELINE config 1 (tests/golden/g_step.npz: both backbones, the fusion encoder, every head, the
    relative-position table, the mask token) at 1e-4 of each gradient's max — the bf16 training path asserts 3e-2 / 1e-1
    on the same fixtures — and the same set of statically unused parameters."""
    from clover_amd import parity
    g = gutil.load('g_step.npz')
    batch = {k: v.to(DEV) for k, v in cf.cf_batch(B, tag=f'step{B}').items()}
    model.zero_grad(set_to_none=True)
    with parity.mode():
        out = model.train_step(batch, None)
        out['loss'].backward()
    named = dict(model.named_parameters())
    worst = {}
    for k in [n[len(f'B{B}.grad.'):-4] for n in g.files if n.startswith(f'B{B}.grad.') and n.endswith('.sub')]:
        sub, stats = gutil.packed(named[k].grad)
        gsub = g[f'B{B}.grad.{k}.sub'].astype(np.float64)
        assert g[f'B{B}.grad.{k}.stats'][2] == stats[2]
        worst[k] = np.abs(sub - gsub).max() / max(np.abs(gsub).max(), 1e-20)
    print('parity grad rel errors', B, worst)
    assert len(worst) == 13
    for k, e in worst.items():
        assert e < 1e-4, (k, e)
    unused = sorted(k for k, p in named.items() if p.grad is None)
    assert unused == gutil.unused_params()
    model.zero_grad(set_to_none=True)


@pytest.mark.parametrize('case', [(2, 4, 14, 14, 48, 3, True), (1, 8, 7, 7, 64, 2, True), (2, 2, 14, 14, 32, 1, False),
                                  (1, 16, 14, 14, 32, 2, True)])
def test_window_attention_f32_backward(case):
    """clv_attn_f32_bwd (mode 1) against fp64 autograd through roll + partition + WindowAttention3D + reverse + un-roll over the
    oracle's index helpers: d qkv and the relative-position-table gradient."""
    from test_kernels_gpu import ref_window_attention
    from clover_amd import ops, parity
    from clover_amd.backbones.swin_transformer_3d import window_geometry
    B, D, H, W, C, nH, shifted = case
    cfg_ws, cfg_ss = (8, 7, 7), ((4, 3, 3) if shifted else (0, 0, 0))
    qkv = rnd(B, D, H, W, 3 * C, seed=71)
    table = rnd((2 * 8 - 1) * 13 * 13, nH, scale=0.5, seed=72)
    dout = rnd(B, D, H, W, C, seed=73)
    q64, t64 = qkv.double().requires_grad_(), table.double().requires_grad_()
    o_ref = ref_window_attention(q64, t64, ix.relative_position_index(cfg_ws), cfg_ws, cfg_ss, nH)
    (o_ref * dout.double()).sum().backward()
    ws, ss, rid = window_geometry((D, H, W), cfg_ws, cfg_ss, DEV)
    qg, tg = qkv.to(DEV).requires_grad_(), table.to(DEV).requires_grad_()
    with parity.mode():
        o = ops.window_attention(qg, tg, rid, ws, ss, nH, table_window=cfg_ws)
    o.backward(dout.to(DEV))
    assert rel(o, o_ref.detach()) < 1e-5
    assert rel(qg.grad, q64.grad) < 2e-5, rel(qg.grad, q64.grad)
    assert rel(tg.grad, t64.grad) < 2e-5, rel(tg.grad, t64.grad)


@pytest.mark.parametrize('B,S,nH,hd', [(3, 16, 2, 64), (2, 228, 12, 64), (2, 40, 4, 32)])
def test_seq_attention_f32_backward(B, S, nH, hd):
    from clover_amd import ops, parity
    Hd = nH * hd
    qkv = rnd(B, S, 3 * Hd, seed=74)
    dout = rnd(B, S, Hd, seed=75)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[0, S - 5:] = 0
    ext = om.extended_mask(mask)
    q64 = qkv.double().requires_grad_()
    q, k, v = q64.view(B, S, 3, nH, hd).permute(2, 0, 3, 1, 4)
    p = (q @ k.transpose(-1, -2) / hd ** 0.5 + ext.double()).softmax(-1)
    o_ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    (o_ref * dout.double()).sum().backward()
    qg = qkv.to(DEV).requires_grad_()
    with parity.mode():
        o = ops.seq_attention(qg, ext.reshape(B, S).to(DEV).contiguous(), nH)
    o.backward(dout.to(DEV))
    assert rel(o, o_ref.detach()) < 1e-5
    assert rel(qg.grad, q64.grad) < 2e-5, rel(qg.grad, q64.grad)
