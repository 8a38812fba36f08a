"""GPU parity of each HIP kernel (through the C ABI) against the oracle / a plain fp32
PyTorch restatement of the same op, on the same seeded inputs.  `-m gpu` only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import indexing as ix          # noqa: E402
from oracle import model as om             # noqa: E402

DEV = 'cuda'
from clover_amd import _lib as _clv_lib  # noqa: E402

BF = _clv_lib.half_dtype()          # the library's 16-bit element type (bf16; fp16 under CLOVER_HALF=f16)


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def ops():
    from clover_amd import ops as o
    return o


# ----------------------------------------------------------------------------- LayerNorm / GELU
@pytest.mark.parametrize('C', [48, 96, 128, 192, 384, 768, 1024, 1536, 3072])
@pytest.mark.parametrize('with_res', [False, True])
def test_layernorm(C, with_res):
    rows = 777
    x = rnd(rows, C, seed=1).to(BF)
    r = rnd(rows, C, seed=2).to(BF) if with_res else None
    g = (1 + 0.1 * rnd(C, seed=3))
    b = 0.1 * rnd(C, seed=4)
    dy = rnd(rows, C, seed=5).to(BF)
    xr = x.float().requires_grad_()
    rr = r.float().requires_grad_() if with_res else None
    gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.layer_norm(xr + rr if with_res else xr, (C,), gr, br, 1e-5)
    yr.backward(dy.float())
    xg = x.to(DEV).requires_grad_()
    rg = r.to(DEV).requires_grad_() if with_res else None
    gg, bg = g.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops().layer_norm(xg, gg, bg, 1e-5, residual=rg)
    y.backward(dy.to(DEV))
    assert rel(y, yr) < 1e-2
    assert rel(xg.grad, xr.grad) < 2e-2
    if with_res:
        assert rel(rg.grad, rr.grad) < 2e-2
    assert rel(gg.grad, gr.grad) < 1e-2
    assert rel(bg.grad, br.grad) < 1e-2


def test_layernorm_eps_1e12_matches_oracle():
    P = {'n.weight': 1 + 0.1 * rnd(128, seed=1), 'n.bias': 0.1 * rnd(128, seed=2)}
    x = rnd(5, 16, 128, seed=3).to(BF)
    yr = om.layer_norm(P, 'n', x.float(), 1e-12)
    y = ops().layer_norm(x.to(DEV), P['n.weight'].to(DEV), P['n.bias'].to(DEV), 1e-12)
    assert rel(y, yr) < 1e-2


def test_gelu():
    x = rnd(1000, 771, scale=2.0, seed=1).to(BF)
    dy = rnd(1000, 771, seed=2).to(BF)
    xr = x.float().requires_grad_()
    yr = om.gelu(xr)
    yr.backward(dy.float())
    xg = x.to(DEV).requires_grad_()
    y = ops().gelu(xg)
    y.backward(dy.to(DEV))
    assert rel(y, yr) < 1e-2
    assert rel(xg.grad, xr.grad) < 1e-2


# ----------------------------------------------------------------------------- window attention
def ref_window_attention(qkv, table, rel_index_full, cfg_ws, cfg_ss, nH):
    """fp32 restatement of roll + partition + WindowAttention3D core + reverse + unroll on a qkv
    tensor in the natural layout (swin_transformer_3d.py:375-397, 459-476), via the oracle's index helpers."""
    B, D, H, W, C3 = qkv.shape
    C = C3 // 3
    hd = C // nH
    ws, ss = ix.get_window_size((D, H, W), cfg_ws, cfg_ss)
    N = ws[0] * ws[1] * ws[2]
    sh = torch.roll(qkv, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3)) if any(ss) else qkv
    xw = om._t_window_partition(sh, ws)                                   # [B_, N, 3C]
    B_ = xw.shape[0]
    q, k, v = xw.reshape(B_, N, 3, nH, hd).permute(2, 0, 3, 1, 4)
    attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
    idx = torch.from_numpy(rel_index_full[:N, :N].reshape(-1).copy())
    bias = table[idx].reshape(N, N, -1).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if any(ss):
        mask = torch.from_numpy(ix.compute_mask(D, H, W, ws, ss))
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, nH, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, nH, N, N)
    attn = attn.softmax(-1)
    o = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    o = om._t_window_reverse(o.view(-1, *(ws + (C,))), ws, B, D, H, W)
    if any(ss):
        o = torch.roll(o, shifts=ss, dims=(1, 2, 3))
    return o


WIN_CASES = [
    # (B, D, H, W, C, nH, block shift on?)
    (2, 4, 14, 14, 96, 3, False),
    (2, 4, 14, 14, 96, 3, True),
    (1, 2, 14, 14, 48, 3, True),      # hd 16, N = 98
    (2, 4, 7, 7, 64, 2, True),        # clamped window, shift zeroed on every axis -> no mask
    (1, 16, 14, 14, 64, 2, True),     # (8,7,7) window, temporal shift 4, N = 392
    (1, 4, 14, 14, 128, 2, True),     # hd 64
]


@pytest.mark.parametrize('bwd_one', ['2', '0'])           # backward: the one-kernel form wherever it exists / dQ + dK/dV kernels
@pytest.mark.parametrize('dbias_index', ['1', '0'])       # table gradient: precomputed offset table / index arithmetic
@pytest.mark.parametrize('case', WIN_CASES)
def test_window_attention(case, dbias_index, bwd_one, monkeypatch):
    from clover_amd.backbones.swin_transformer_3d import window_geometry
    monkeypatch.setenv('CLOVER_DBIAS_INDEX', dbias_index)
    monkeypatch.setenv('CLV_ATTN_BWD_ONE', bwd_one)
    B, D, H, W, C, nH, shifted = case
    cfg_ws, cfg_ss = (8, 7, 7), ((4, 3, 3) if shifted else (0, 0, 0))
    qkv = rnd(B, D, H, W, 3 * C, seed=11).to(BF)
    table = rnd((2 * 8 - 1) * 13 * 13, nH, scale=0.5, seed=12)
    do = rnd(B, D, H, W, C, seed=13).to(BF)
    rpi = ix.relative_position_index(cfg_ws)
    qr = qkv.float().requires_grad_()
    tr = table.clone().requires_grad_()
    o_ref = ref_window_attention(qr, tr, rpi, cfg_ws, cfg_ss, nH)
    o_ref.backward(do.float())

    ws, ss, rid = window_geometry((D, H, W), cfg_ws, cfg_ss, DEV)
    qg = qkv.to(DEV).requires_grad_()
    tg = table.to(DEV).requires_grad_()
    o = ops().window_attention(qg, tg, rid, ws, ss, nH, table_window=cfg_ws)
    o.backward(do.to(DEV))
    assert rel(o, o_ref) < 2e-2, rel(o, o_ref)
    assert rel(qg.grad, qr.grad) < 3e-2, rel(qg.grad, qr.grad)
    assert rel(tg.grad, tr.grad) < 3e-2, rel(tg.grad, tr.grad)


@pytest.mark.parametrize('B,D,H,W,C', [(2, 4, 56, 56, 96), (2, 2, 28, 28, 192), (3, 4, 14, 14, 384), (1, 1, 2, 2, 8),
                                       (2, 3, 6, 10, 128)])
@pytest.mark.parametrize('with_res,with_scale', [(True, True), (True, False), (False, False), (False, True)])
def test_merge_layer_norm(B, D, H, W, C, with_res, with_scale):
    """PatchMerging gather + LayerNorm (+ pending residual, + DropPath factor) in one kernel each way, against
    the oracle's gather (swin_transformer_3d.py:531-539 order) followed by torch LayerNorm."""
    x = rnd(B, D, H, W, C, seed=141).to(BF)
    r = rnd(B, D, H, W, C, seed=142).to(BF) if with_res else None
    sc = torch.tensor([0.0, 1.25, 1.25][:B] if B <= 3 else [1.25] * B) if with_scale else None
    g = 1 + 0.1 * rnd(4 * C, seed=143)
    b = 0.1 * rnd(4 * C, seed=144)
    dy = rnd(B, D, H // 2, W // 2, 4 * C, seed=145).to(BF)

    xr = x.float().requires_grad_()
    rr = r.float().requires_grad_() if with_res else None
    gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
    t = xr * (sc.view(B, 1, 1, 1, 1) if with_scale else 1.0)
    if with_res:
        t = t + rr
    gathered = torch.cat([t[:, :, 0::2, 0::2], t[:, :, 1::2, 0::2], t[:, :, 0::2, 1::2], t[:, :, 1::2, 1::2]], -1)
    assert np.array_equal(gathered.detach().numpy(), ix.patch_merging_gather(t.detach().numpy()))   # the oracle's order
    y_ref = F.layer_norm(gathered, (4 * C,), gr, br, 1e-5)
    y_ref.backward(dy.float())

    xg = x.to(DEV).requires_grad_()
    rg = r.to(DEV).requires_grad_() if with_res else None
    gg, bg = g.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops().merge_layer_norm(xg, gg, bg, 1e-5, residual=rg, x_scale=sc.to(DEV) if with_scale else None)
    y.backward(dy.to(DEV))
    assert y.shape == y_ref.shape
    assert rel(y, y_ref) < 2e-2, rel(y, y_ref)
    assert rel(xg.grad, xr.grad) < 2e-2, rel(xg.grad, xr.grad)
    if with_res:
        assert rel(rg.grad, rr.grad) < 2e-2
    assert rel(gg.grad, gr.grad) < 2e-2 and rel(bg.grad, br.grad) < 2e-2
    if with_scale:
        assert xg.grad[0].abs().max().item() == 0.0          # dropped path: no gradient into the branch


def test_window_attention_deferred_table_gradient():
    """ops.defer_folds(): the table-gradient gathers of several window-attention blocks (different geometries, one table
    used twice) are left pending by their backward calls and run as batched launches when the segment closes
    (clv_attn_dbias_gather_batch); the sinks end up with what the immediate gathers give."""
    from clover_amd.backbones.swin_transformer_3d import window_geometry
    cases = [(2, 4, 14, 14, 64, 2, True), (1, 8, 14, 14, 96, 3, False), (2, 4, 7, 7, 64, 2, False)]
    inputs = []
    for i, (B, D, H, W, Cc, nH, shifted) in enumerate(cases):
        ws, ss, rid = window_geometry((D, H, W), (8, 7, 7), (4, 3, 3) if shifted else (0, 0, 0), DEV)
        inputs.append((rnd(B, D, H, W, 3 * Cc, seed=610 + i).to(BF).to(DEV), rnd(15 * 13 * 13, nH, scale=0.5, seed=620 + i).to(DEV),
                       rnd(B, D, H, W, Cc, seed=630 + i).to(BF).to(DEV), rid, ws, ss, nH))

    def run(deferred):
        sinks, grads = [], []
        tabs = []
        for qkv, table, do, rid, ws, ss, nH in inputs:
            t = table.clone().requires_grad_()
            t._clv_grad = torch.zeros_like(table)              # an engine-style gradient sink: accumulated in place
            t._clv_ready = lambda: None
            tabs.append(t)
            sinks.append(t._clv_grad)

        def body():
            for (qkv, table, do, rid, ws, ss, nH), t in zip(inputs, tabs):
                for rep in range(2):                           # every table takes part in two blocks of the segment
                    q = qkv.clone().requires_grad_()
                    ops().window_attention(q, t, rid, ws, ss, nH, table_window=(8, 7, 7)).backward(do)
                    grads.append(q.grad)
        if deferred:
            with ops().defer_folds():
                body()
                assert len(ops().DBIAS_DEFER) == 2 * len(inputs)
                # what stays alive until the gather is the partial-sum buffer alone, not the block's dS scratch
                # (clv_attn_bwd_work_bytes: GBs per block at 32 frames — 72 GB over config 5's 24 blocks)
                held = sum(item[1].numel() for item in ops().DBIAS_DEFER)
                assert held < 40 * 2 ** 20, held
        else:
            body()
        torch.cuda.synchronize()
        return sinks, grads
    s0, g0 = run(False)
    s1, g1 = run(True)
    for a, b in zip(s0, s1):
        assert float(a.abs().max()) > 0 and rel(b, a) < 1e-5, rel(b, a)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)


@pytest.mark.parametrize('D,N', [(4, 196), (16, 392)])
def test_window_attention_one_kernel_stage_masks(D, N, monkeypatch):
    """clv_attn_bwd on the one-kernel geometries: `clv_attn_bwd_one_kernel` reports them, the stage masks 5 (dQ / dK / dV
    kernel) + 2 (table gradient) together give what one call with stages = 0 gives, and both agree with the two-kernel path."""
    import ctypes as C
    from clover_amd import _lib
    from clover_amd._lib import ClvAttnGeom
    from clover_amd.backbones.swin_transformer_3d import window_geometry
    L = _lib.lib()
    B, H, W, Cc, nH = 2, 14, 14, 64, 2
    ws, ss, rid = window_geometry((D, H, W), (8, 7, 7), (4, 3, 3), DEV)
    assert ws[0] * ws[1] * ws[2] == N
    nW = (D // ws[0]) * (H // ws[1]) * (W // ws[2])
    hd = Cc // nH
    g = ClvAttnGeom(mode=1, groups=B * nW, N=N, nH=nH, hd=hd, D=D, H=H, W=W, wd=ws[0], wh=ws[1], ww=ws[2], sd=ss[0], sh=ss[1],
                    sw=ss[2], ldq=3 * Cc, ldk=3 * Cc, ldv=3 * Cc, ldo=Cc, bwd=8, bwh=7, bww=7, scale=hd ** -0.5, dropout_p=0.0)
    qkv = rnd(B, D, H, W, 3 * Cc, seed=31).to(BF).to(DEV)
    table = rnd(15 * 13 * 13, nH, scale=0.5, seed=32).to(DEV)
    do = rnd(B, D, H, W, Cc, seed=33).to(BF).to(DEV)
    o = torch.empty(B, D, H, W, Cc, device=DEV, dtype=BF)
    lse = torch.empty(g.groups * nH * N, device=DEV)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = qkv.data_ptr()
    assert L.clv_attn_fwd(C.c_void_p(p), C.c_void_p(p + 2 * Cc), C.c_void_p(p + 4 * Cc), P(o), P(lse), P(table), P(rid), None, None,
                          C.byref(g), st) == 0
    g.dbias_index = ops()._dbias_index(g, qkv.device)
    work = torch.empty(L.clv_attn_bwd_work_bytes(C.byref(g)), device=DEV, dtype=torch.uint8)
    dsum = torch.empty_like(lse)

    def run(masks):
        dqkv = torch.zeros_like(qkv)
        dtab = torch.zeros_like(table)
        d = dqkv.data_ptr()
        for m in masks:
            rc = L.clv_attn_bwd(C.c_void_p(p), C.c_void_p(p + 2 * Cc), C.c_void_p(p + 4 * Cc), P(o), P(do), P(lse), P(table), P(rid),
                                None, C.c_void_p(d), C.c_void_p(d + 2 * Cc), C.c_void_p(d + 4 * Cc), P(dtab), P(dsum), P(work), None, m,
                                C.byref(g), st)
            assert rc == 0, (m, rc)
        torch.cuda.synchronize()
        return dqkv.float().cpu(), dtab.cpu()
    monkeypatch.setenv('CLV_ATTN_BWD_ONE', '2')
    assert L.clv_attn_bwd_one_kernel(C.byref(g)) == 1
    a_q, a_t = run([0])
    b_q, b_t = run([5, 2])
    assert torch.equal(a_q, b_q) and rel(a_t, b_t) < 1e-6
    monkeypatch.setenv('CLV_ATTN_BWD_ONE', '0')
    assert L.clv_attn_bwd_one_kernel(C.byref(g)) == 0
    c_q, c_t = run([0])
    assert rel(a_q, c_q) < 2e-2 and rel(a_t, c_t) < 2e-2, (rel(a_q, c_q), rel(a_t, c_t))


# ----------------------------------------------------------------------------- sequence attention
@pytest.mark.parametrize('B,S,nH,hd', [(3, 16, 2, 64), (2, 32, 12, 64), (2, 228, 12, 64), (2, 40, 4, 32), (1, 408, 2, 64)])
def test_seq_attention(B, S, nH, hd):
    Hd = nH * hd
    qkv = rnd(B, S, 3 * Hd, seed=21).to(BF)
    do = rnd(B, S, Hd, seed=22).to(BF)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[0, S - 5:] = 0
    ext = om.extended_mask(mask)                                           # [B,1,1,S]
    qr = qkv.float().requires_grad_()
    q, k, v = qr.view(B, S, 3, nH, hd).permute(2, 0, 3, 1, 4)
    p = (q @ k.transpose(-1, -2) / hd ** 0.5 + ext).softmax(-1)
    o_ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    o_ref.backward(do.float())
    qg = qkv.to(DEV).requires_grad_()
    o = ops().seq_attention(qg, ext.reshape(B, S).to(DEV).contiguous(), nH)
    o.backward(do.to(DEV))
    assert rel(o, o_ref) < 2e-2, rel(o, o_ref)
    assert rel(qg.grad, qr.grad) < 3e-2, rel(qg.grad, qr.grad)


# ----------------------------------------------------------------------------- patch embed
@pytest.mark.parametrize('C,B,T,HW', [(48, 2, 4, 112), (96, 2, 8, 56), (128, 1, 4, 56)])
def test_patch_embed(C, B, T, HW):
    x = rnd(B, 3, T, HW, HW, seed=31)
    P = {'proj.weight': rnd(C, 3, 2, 4, 4, scale=0.1, seed=32), 'proj.bias': 0.1 * rnd(C, seed=33),
         'norm.weight': 1 + 0.1 * rnd(C, seed=34), 'norm.bias': 0.1 * rnd(C, seed=35)}
    mt = rnd(1, C, 1, 1, 1, scale=0.3, seed=36)
    vm = (rnd(B, 1, 7, 7, seed=37) > 0.5).long()
    Pr = {k: v.clone().requires_grad_() for k, v in P.items()}
    mtr = mt.clone().requires_grad_()
    cfg = dict(patch_size=(2, 4, 4), embed_dim=C, patch_norm=True)
    y = om.patch_embed(Pr, '', x, cfg)                                      # [B,C,T',H',W']
    Tp, Hp = y.shape[2], y.shape[3]
    w = torch.from_numpy(ix.mask_blend_weight(vm.numpy(), Tp, Hp, Hp)).float()
    ym = y * (1. - w) + mtr.expand(B, -1, Tp, Hp, Hp) * w
    g1 = rnd(*y.shape, seed=38).to(BF).float()
    g2 = rnd(*y.shape, seed=39).to(BF).float()
    ((y * g1).sum() + (ym * g2).sum()).backward()

    Pg = {k: v.to(DEV).requires_grad_() for k, v in P.items()}
    mtg = mt.to(DEV).requires_grad_()
    clean, masked = ops().patch_embed(x.to(DEV), Pg['proj.weight'], Pg['proj.bias'], Pg['norm.weight'],
                                      Pg['norm.bias'], mtg, vm.to(DEV))
    assert rel(clean.permute(0, 4, 1, 2, 3), y) < 2e-2
    assert rel(masked.permute(0, 4, 1, 2, 3), ym) < 2e-2
    ((clean.float() * g1.to(DEV).permute(0, 2, 3, 4, 1)).sum() + (masked.float() * g2.to(DEV).permute(0, 2, 3, 4, 1)).sum()).backward()
    for k in P:
        assert rel(Pg[k].grad, Pr[k].grad) < 3e-2, (k, rel(Pg[k].grad, Pr[k].grad))
    assert rel(mtg.grad, mtr.grad) < 2e-2


# ----------------------------------------------------------------------------- losses
@pytest.mark.parametrize('V', [30522, 1000, 1001])      # the BERT vocabulary; fewer pairs than threads; odd: 2-byte fallback
@pytest.mark.parametrize('dtype', [torch.float32, BF])
def test_focal_ce(dtype, V):
    R = 37
    logits = rnd(R, V, scale=2.0, seed=41).to(dtype).clone()
    labels = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(42))
    labels[::3] = -100
    rows = torch.where(labels != -100)[0]
    lr = logits.detach().float().clone().requires_grad_()
    loss_ref = om.focal_loss_multiclass(lr[rows], labels[rows], 2.0)
    loss_ref.backward()
    lg = logits.detach().to(DEV).requires_grad_()
    loss = ops().focal_ce_masked(lg, labels.to(DEV), 2.0)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4 * max(1, abs(loss_ref.item()))
    assert rel(lg.grad, lr.grad) < (1e-4 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('G', [1, 2, 8, 64, 200])
def test_infonce(G):
    Dm = 768 if G > 2 else 128
    es = [rnd(G, Dm, seed=50 + k) for k in range(4)]
    # make positives close so that the ranking hinge is active for some rows and not others
    es[1] = es[0] * 0.7 + es[1] * 0.5
    es[2] = es[0] * 0.6 + es[2] * 0.6
    er = [e.clone().requires_grad_() for e in es]
    l = om.exclusive_nce_rank_loss(*er, temperature=0.05, margin=5.0, gather=False)
    (l['nce_loss'] * 1.3 + l['rank_t_tm_loss'] * 0.7).backward()
    eg = [e.to(DEV).requires_grad_() for e in es]
    nce, rank = ops().exclusive_infonce_rank(*eg, 0.05, 5.0)
    (nce * 1.3 + rank * 0.7).backward()
    assert abs(nce.item() - l['nce_loss'].item()) < 2e-4 * max(1, abs(l['nce_loss'].item())), (nce.item(), l['nce_loss'].item())
    assert abs(rank.item() - l['rank_t_tm_loss'].item()) < 2e-4 * max(1, abs(l['rank_t_tm_loss'].item()))
    for k in range(4):
        if er[k].grad.abs().max() > 0:
            assert rel(eg[k].grad, er[k].grad) < 2e-3, (k, rel(eg[k].grad, er[k].grad))


@pytest.mark.parametrize('G', [2, 8, 64])
def test_infonce_packed_slots(G):
    """The packed entry (four slots of one gathered [G, 6, Dm] tensor, strided reads / strided gradient writes)
    equals the four-tensor entry, for both slot orders the step uses."""
    Dm = 768
    packed = rnd(G, 6, Dm, seed=81)
    packed[:, 1] = packed[:, 0] * 0.7 + packed[:, 1] * 0.5
    for slots in [(0, 1, 2, 3), (1, 0, 4, 5)]:
        pg = packed.to(DEV).requires_grad_()
        nce, rank = ops().exclusive_infonce_rank_packed(pg, slots, 0.05, 5.0)
        (nce * 1.3 + rank * 0.7).backward()
        es = [packed[:, s].clone().to(DEV).requires_grad_() for s in slots]
        nce2, rank2 = ops().exclusive_infonce_rank(*es, 0.05, 5.0)
        (nce2 * 1.3 + rank2 * 0.7).backward()
        assert nce.item() == nce2.item() and rank.item() == rank2.item()
        for k, s_ in enumerate(slots):
            assert torch.equal(pg.grad[:, s_], es[k].grad)
        unused = [u for u in range(6) if u not in slots]
        assert pg.grad[:, unused].abs().max().item() == 0.0


@pytest.mark.parametrize('G', [1, 2, 8, 64, 200])
def test_infonce_pair_equals_two_packed_evaluations(G):
    """clv_infonce_pair_fwd / _bwd (both evaluations of the step in the same launches, the whole packed gradient written by
    the backward) against two calls of the packed entry: the four losses bit for bit; the gradient bit for bit on the slots
    one evaluation reads, to rounding (the two contributions are summed before the normalisation's Jacobian instead of after)
    on the two slots both read; a slot nobody reads gets zeros.  Distinct upstream gradients per loss."""
    Dm, k = 768, 7
    packed = rnd(G, k, Dm, seed=181)
    packed[:, 1] = packed[:, 0] * 0.7 + packed[:, 1] * 0.5
    packed[:, 4] = packed[:, 1] * 0.6 + packed[:, 4] * 0.6
    sa, sb = (0, 1, 2, 3), (1, 0, 4, 5)
    wts = (1.3, 0.7, 0.9, 1.1)
    pg = packed.to(DEV).requires_grad_()
    outs = ops().exclusive_infonce_rank_pair(pg, sa, sb, 0.05, 5.0)
    sum(w * o for w, o in zip(wts, outs)).backward()
    pr = packed.to(DEV).requires_grad_()
    na, ra = ops().exclusive_infonce_rank_packed(pr, sa, 0.05, 5.0)
    nb, rb = ops().exclusive_infonce_rank_packed(pr, sb, 0.05, 5.0)
    (wts[0] * na + wts[1] * ra + wts[2] * nb + wts[3] * rb).backward()
    for o, r in zip(outs, (na, ra, nb, rb)):
        assert o.item() == r.item(), (o.item(), r.item())
    for s_ in (2, 3, 4, 5):
        assert torch.equal(pg.grad[:, s_], pr.grad[:, s_]), s_
    for s_ in (0, 1):
        assert rel(pg.grad[:, s_], pr.grad[:, s_]) < 1e-5, (s_, rel(pg.grad[:, s_], pr.grad[:, s_]))
    assert pg.grad[:, 6].abs().max().item() == 0.0
    import ctypes as C
    from clover_amd import _lib
    L = _lib.lib()
    bad = (C.c_int32 * 8)(0, 1, 2, 2, 1, 0, 4, 5)              # a repeated slot inside one evaluation
    out = torch.empty(4, device=DEV)
    work = torch.empty(2 * L.clv_infonce_work_floats(G, Dm), device=DEV)
    assert L.clv_infonce_pair_fwd(pg.data_ptr(), bad, out.data_ptr(), work.data_ptr(), G, k, Dm, 0.05, 5.0, None) != 0


def test_norm_softmax_loss_goldens():
    """clv_normsoftmax_fwd/bwd against the reference's own NormSoftmaxLoss numbers (g_finetune.npz): both norm
    clamps, ragged widths, a zero-norm row, the sim_mat entry."""
    import closed_form as cf
    import gutil
    g = gutil.load('g_finetune.npz')
    for cos in (False, True):
        for G in (1, 2, 4, 8, 33):
            for Dm in (128, 50):
                tag = f'loss.cos{int(cos)}.G{G}.D{Dm}'
                v = cf.cf_float(tag + '.v', (G, Dm), 1.3).to(DEV).requires_grad_()
                t = cf.cf_float(tag + '.t', (G, Dm), 0.9, 0.1).to(DEV).requires_grad_()
                loss = ops().norm_softmax_loss(v, t, temperature=0.05 if cos else 0.07, eps=1e-8 if cos else 1e-12)
                loss.backward()
                ref = float(g[tag])
                assert abs(loss.item() - ref) < 2e-4 * max(1, abs(ref)), (tag, loss.item(), ref)
                assert rel(v.grad, torch.from_numpy(g[tag + '.dv'])) < 2e-3, tag
                assert rel(t.grad, torch.from_numpy(g[tag + '.dt'])) < 2e-3, tag
        v = cf.cf_float('loss.zero.v', (4, 32), 1.0)
        v[2] = 0
        t = cf.cf_float('loss.zero.t', (4, 32), 1.0).to(DEV).requires_grad_()
        vg = v.to(DEV).requires_grad_()
        loss = ops().norm_softmax_loss(vg, t, temperature=0.07, eps=1e-8 if cos else 1e-12)
        loss.backward()
        assert abs(loss.item() - float(g[f'loss.zero.cos{int(cos)}'])) < 2e-4
        assert rel(t.grad, torch.from_numpy(g[f'loss.zero.cos{int(cos)}.dt'])) < 2e-3
        assert torch.isfinite(vg.grad).all()
    x = cf.cf_float('loss.sim', (6, 6), 9.0).to(DEV).requires_grad_()
    loss = ops().norm_softmax_loss(sim_mat=x)
    loss.backward()
    assert abs(loss.item() - float(g['loss.sim'])) < 2e-4 * abs(float(g['loss.sim']))
    assert rel(x.grad, torch.from_numpy(g['loss.sim.dx'])) < 2e-3


@pytest.mark.parametrize('G,Dm', [(128, 768), (1000, 768)])
def test_norm_softmax_loss_large(G, Dm):
    """Fine-tuning sizes (16 videos x 8 ranks; a 1000-pair MSRVTT evaluation split) against the oracle."""
    v, t = rnd(G, Dm, seed=71), rnd(G, Dm, seed=72)
    t = 0.5 * v + t
    vr, tr = v.clone().requires_grad_(), t.clone().requires_grad_()
    lref = om.norm_softmax_loss(vr, tr, temperature=0.05, cos_sim=True, gather=False)
    lref.backward()
    vg, tg = v.to(DEV).requires_grad_(), t.to(DEV).requires_grad_()
    loss = ops().norm_softmax_loss(vg, tg, temperature=0.05, eps=1e-8)
    loss.backward()
    assert abs(loss.item() - lref.item()) < 2e-4 * max(1, abs(lref.item())), (loss.item(), lref.item())
    assert rel(vg.grad, vr.grad) < 2e-3 and rel(tg.grad, tr.grad) < 2e-3


# ----------------------------------------------------------------------------- optimizer
def test_adamw_and_clip():
    n = 100003
    p0 = rnd(n, seed=61)
    grads = [rnd(n, scale=s, seed=62 + i) for i, s in enumerate([1.0, 30.0, 0.1])]
    pr = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pr], lr=1e-2, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.005)
    pg = p0.clone().to(DEV)
    pad = (-n) % 4
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    sh = torch.zeros(n + pad, device=DEV, dtype=BF)
    for step, g in enumerate(grads, 1):
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 15.0)
        opt.step()
        gg = g.to(DEV)
        acc = torch.zeros(1, device=DEV)
        ops().sumsq_accumulate(gg, acc)
        assert abs(acc.item() - (g.double() ** 2).sum().item()) < 1e-3 * (g.double() ** 2).sum().item()
        ops().adamw_step(pg, gg, m, v, sh, acc, 1e-2, 0.9, 0.98, 1e-8, 0.005, step, 15.0)
    assert rel(pg, pr.data) < 1e-5
    assert rel(sh[:n].float(), pr.data) < 1e-2
    # non-finite grad norm -> step skipped
    before = pg.clone()
    acc = torch.full((1,), float('inf'), device=DEV)
    ops().adamw_step(pg, grads[0].to(DEV), m, v, sh, acc, 1e-2, 0.9, 0.98, 1e-8, 0.005, 4, 15.0)
    assert torch.equal(before, pg)


def test_bf16_gradient_variants_of_norm_and_adamw():
    """clv_pack_bf16 + clv_sumsq_bf16 + clv_adamw_step_dev_bf16g (the data-parallel path: gradients travel and are read as
    bf16, the update stays fp32) against the fp32 kernels fed the SAME bf16-rounded gradient: the same update to fp32 round-off, and
    the norm equal to the fp32 kernel's on the rounded values."""
    n = 1_000_003
    p0, g = rnd(n, seed=301), rnd(n, seed=302) * 0.01
    gb = torch.empty(n, dtype=BF, device=DEV)
    ops().pack_bf16(g.to(DEV), gb)
    assert torch.equal(gb.cpu(), g.to(BF))
    gr = gb.float()                                            # what the fp32 kernels see for comparison
    out = {}
    for tag, grad in (('f32', gr), ('bf16', gb)):
        p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        sh = torch.zeros(n, device=DEV, dtype=BF)
        acc, st = torch.zeros(1, device=DEV), ops().optim_state_new(DEV)
        for _ in range(2):
            ops().sumsq_accumulate(grad, acc)
            ops().optim_prep(acc, st, 0.9, 0.98, 15.0, 0.5)
            ops().adamw_step_dev(p, grad, m, v, sh, st, 1e-3, 0.9, 0.98, 1e-8, 0.005)
        out[tag] = (p.cpu(), m.cpu(), v.cpu(), sh.cpu(), ops().optim_state_read(st)['norm'])
    # same arithmetic on the same values; the clip coefficient comes from an atomically accumulated norm (summation order
    # differs between launches), so equality holds to fp32 round-off, and exactly for the bf16 shadow except at ties
    for a, b in zip(out['f32'][:3], out['bf16'][:3]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-9)
    assert (out['f32'][3] != out['bf16'][3]).float().mean().item() < 1e-3
    assert abs(out['f32'][4] - out['bf16'][4]) <= 1e-6 * out['f32'][4]


def test_adamw_device_state_two_option_groups_and_skip():
    """clv_optim_prep + clv_adamw_step_dev (the engine's path): two slabs with different (weight_decay, lr) — the
    per-parameter options of a paramwise_cfg — against torch.optim.AdamW param groups with a global-norm clip; a
    non-finite step is skipped on the device and leaves Adam's count (bias corrections) where it was."""
    na, nb = 50001, 30002
    pa0, pb0 = rnd(na, seed=161), rnd(nb, seed=162)
    pra, prb = torch.nn.Parameter(pa0.clone()), torch.nn.Parameter(pb0.clone())
    opt = torch.optim.AdamW([dict(params=[pra], weight_decay=0.5, lr=1e-2), dict(params=[prb], weight_decay=0.25, lr=2.5e-3)],
                            betas=(0.9, 0.98), eps=1e-8)
    pga, pgb = pa0.clone().to(DEV), pb0.clone().to(DEV)
    ma, va, mb, vb = (torch.zeros(n, device=DEV) for n in (na, na, nb, nb))
    sha = torch.zeros(na + (-na) % 4, device=DEV, dtype=BF)
    state = ops().optim_state_new(DEV)
    acc = torch.zeros(1, device=DEV)

    def dev_step(ga, gb, world=1):
        ops().sumsq_accumulate(ga, acc)
        ops().sumsq_accumulate(gb, acc)
        ops().optim_prep(acc, state, 0.9, 0.98, 15.0, 1.0 / world)
        ops().adamw_step_dev(pga, ga, ma, va, sha, state, 1e-2, 0.9, 0.98, 1e-8, 0.5)
        ops().adamw_step_dev(pgb, gb, mb, vb, None, state, 2.5e-3, 0.9, 0.98, 1e-8, 0.25)
    for i, sc in enumerate([1.0, 30.0, 0.1]):
        ga, gb = rnd(na, scale=sc, seed=170 + i), rnd(nb, scale=sc, seed=180 + i)
        pra.grad, prb.grad = ga.clone(), gb.clone()
        torch.nn.utils.clip_grad_norm_([pra, prb], 15.0)
        opt.step()
        if i == 1:                                          # a non-finite step in between changes nothing
            bad = ga.clone()
            bad[3] = float('inf')
            before = (pga.clone(), ma.clone(), pgb.clone())
            dev_step(bad.to(DEV), gb.to(DEV))
            assert torch.equal(before[0], pga) and torch.equal(before[1], ma) and torch.equal(before[2], pgb)
            assert ops().optim_state_read(state)['skip'] == 1
        dev_step(2.0 * ga.to(DEV), 2.0 * gb.to(DEV), world=2)       # grad_scale = 1/W undoes the summed gradients
        assert float(acc.item()) == 0.0                     # prep re-zeroes the accumulator
    st = ops().optim_state_read(state)
    assert st['t'] == 3 and st['skipped'] == 1 and st['skip'] == 0
    assert rel(pga, pra.data) < 1e-5 and rel(pgb, prb.data) < 1e-5
    assert rel(sha[:na].float(), pra.data) < 1e-2
    gn = (rnd(na, scale=0.1, seed=172).double().pow(2).sum() + rnd(nb, scale=0.1, seed=182).double().pow(2).sum()).sqrt()
    assert abs(st['norm'] - gn.item()) < 1e-3 * gn.item()


# ----------------------------------------------------------------------------- LDS-tiled GEMM + epilogues
@pytest.mark.parametrize('M,N,K', [(12544, 1536, 384), (3136, 768, 3072), (3648, 2304, 768), (50176, 192, 192),
                                   (1000, 576, 192), (130, 64, 64), (257, 200, 128), (4096, 1152, 384),
                                   (48, 512, 128), (17, 128, 512), (1, 64, 64),                  # fewer rows than a tile
                                   (1568, 4096, 1024), (784, 4104, 1024), (1568, 1024, 4096)])   # N > 3072: bias from global memory (Swin-B stage 3)
def test_gemm_nt_epilogues(M, N, K):
    """clv_gemm_nt against fp32 torch on the same bf16 operands: plain, bias, bias + GELU (pre-activation kept) and
    the GELU-backward epilogue; ragged M / N edges, both tile widths."""
    F_ = torch.nn.functional
    a = rnd(M, K, seed=301).to(BF)
    w = rnd(N, K, scale=0.05, seed=302).to(BF)
    bias = rnd(N, scale=0.2, seed=303)
    ref = a.float() @ w.float().t()
    ag, wg, bg = a.to(DEV), w.to(DEV), bias.to(DEV)
    scale = ref.abs().max().item()
    c0 = ops().gemm_nt(ag, wg, epilogue=ops().GEMM_EPI_NONE)
    assert (c0.float().cpu() - ref).abs().max().item() < 6e-3 * scale
    c1 = ops().gemm_nt(ag, wg, bg, epilogue=ops().GEMM_EPI_BIAS)
    assert (c1.float().cpu() - (ref + bias)).abs().max().item() < 6e-3 * scale
    act, pre = ops().gemm_nt(ag, wg, bg, epilogue=ops().GEMM_EPI_BIAS_GELU)
    assert (pre.float().cpu() - (ref + bias)).abs().max().item() < 6e-3 * scale
    assert (act.float().cpu() - F_.gelu(ref + bias)).abs().max().item() < 6e-3 * scale
    prein = rnd(M, N, seed=304).to(BF)
    x = prein.float().requires_grad_()
    F_.gelu(x).backward(torch.ones_like(x))
    c3 = ops().gemm_nt(ag, wg, aux=prein.to(DEV), epilogue=ops().GEMM_EPI_DGELU)
    want = ref.to(BF).float() * x.grad                    # the unfused path rounds d act to bf16 first
    assert (c3.float().cpu() - want).abs().max().item() < 8e-3 * scale
    # strided A (a column slice of a wider tensor, as the q|k|v views are); 32 elements keep 16-byte alignment
    wide = torch.zeros(M, K + 64, dtype=BF)
    wide[:, 32:32 + K] = a
    c4 = ops().gemm_nt(wide.to(DEV)[:, 32:32 + K], wg, epilogue=ops().GEMM_EPI_NONE)
    assert torch.equal(c4, c0)


@pytest.mark.parametrize('M,N,K', [(3136, 768, 3072), (512, 768, 3072), (3648, 768, 3072), (300, 200, 1536), (1000, 136, 2048)])
@pytest.mark.parametrize('splitk,rot', [('0', '1'), ('2', '0'), ('5', '1'), ('8', '1'), ('1', '1')])
def test_gemm_nt_split_k_and_rotation(M, N, K, splitk, rot, monkeypatch):
    """Few-tile long-contraction shapes: the contraction cut into K slices (fp32 partial slabs + the reduce kernel that
    applies the epilogue) and the per-tile rotation of the K walk — every epilogue against fp32 torch and against the
    one-pass launch (same operands; only the fp32 summation order differs).  splitk '0' = the planner's own choice,
    'n' = forced n slices, '1' = off."""
    F_ = torch.nn.functional
    L = ops()
    import clover_amd._lib as lib
    a = rnd(M, K, seed=311).to(BF)
    w = rnd(N, K, scale=0.05, seed=312).to(BF)
    bias = rnd(N, scale=0.2, seed=313)
    ref = a.float() @ w.float().t()
    scale = ref.abs().max().item()
    ag, wg, bg = a.to(DEV), w.to(DEV), bias.to(DEV)
    prein = rnd(M, N, seed=314).to(BF)
    monkeypatch.setenv('CLV_GEMM_SPLITK', '1')
    monkeypatch.setenv('CLV_GEMM_ROT', '0')
    assert lib.lib().clv_gemm_nt_work_bytes(M, N, K) == 0
    base = dict(none=L.gemm_nt(ag, wg, epilogue=L.GEMM_EPI_NONE), bias=L.gemm_nt(ag, wg, bg, epilogue=L.GEMM_EPI_BIAS),
                gelu=L.gemm_nt(ag, wg, bg, epilogue=L.GEMM_EPI_BIAS_GELU), gelud=L.gemm_nt(ag, wg, bg, epilogue=L.GEMM_EPI_BIAS_GELU_D),
                dgelu=L.gemm_nt(ag, wg, aux=prein.to(DEV), epilogue=L.GEMM_EPI_DGELU),
                mul=L.gemm_nt(ag, wg, aux=prein.to(DEV), epilogue=L.GEMM_EPI_MUL))
    monkeypatch.setenv('CLV_GEMM_SPLITK', splitk)
    monkeypatch.setenv('CLV_GEMM_ROT', rot)
    wb = lib.lib().clv_gemm_nt_work_bytes(M, N, K)
    if splitk not in ('0', '1'):
        assert wb == int(splitk) * M * N * 4, (wb, splitk)
    if splitk == '1':
        assert wb == 0
    got = dict(none=L.gemm_nt(ag, wg, epilogue=L.GEMM_EPI_NONE), bias=L.gemm_nt(ag, wg, bg, epilogue=L.GEMM_EPI_BIAS),
               gelu=L.gemm_nt(ag, wg, bg, epilogue=L.GEMM_EPI_BIAS_GELU), gelud=L.gemm_nt(ag, wg, bg, epilogue=L.GEMM_EPI_BIAS_GELU_D),
               dgelu=L.gemm_nt(ag, wg, aux=prein.to(DEV), epilogue=L.GEMM_EPI_DGELU),
               mul=L.gemm_nt(ag, wg, aux=prein.to(DEV), epilogue=L.GEMM_EPI_MUL))
    assert (got['none'].float().cpu() - ref).abs().max().item() < 6e-3 * scale
    assert (got['bias'].float().cpu() - (ref + bias)).abs().max().item() < 6e-3 * scale
    assert (got['gelu'][0].float().cpu() - F_.gelu(ref + bias)).abs().max().item() < 6e-3 * scale
    for k in base:
        for x, y in zip(got[k] if isinstance(got[k], tuple) else (got[k],), base[k] if isinstance(base[k], tuple) else (base[k],)):
            # one bf16 ulp of the largest value: the two launches round the same fp32 sums taken in a different order
            assert (x.float() - y.float()).abs().max().item() <= 8e-3 * max(1.0, y.float().abs().max().item()), k


@pytest.mark.usefixtures('strict_own_gemm')
@pytest.mark.parametrize('M,C,Hd', [(8192, 192, 768), (12544, 384, 1536), (9000, 128, 512), (1568, 1024, 4096)])
def test_linear_and_mlp_on_the_hip_gemm(M, C, Hd):
    """ops.linear / ops.mlp_gelu on shapes that take clv_gemm_nt (own_gemm_ok): forward, input gradient (through the
    transposed-weight operand) and parameter gradients against fp32 torch on the same bf16 operands."""
    F_ = torch.nn.functional
    x = rnd(M, C, seed=321).to(BF)
    w1, b1 = rnd(Hd, C, scale=0.05, seed=322), rnd(Hd, scale=0.1, seed=323)
    w2, b2 = rnd(C, Hd, scale=0.05, seed=324), rnd(C, scale=0.1, seed=325)
    dy = rnd(M, C, seed=326).to(BF)
    xr = x.float().requires_grad_()
    pr = [t.clone().requires_grad_() for t in (w1, b1, w2, b2)]
    h = F_.linear(xr, pr[0].to(BF).float(), pr[1])
    yr = F_.linear(F_.gelu(h).to(BF).float(), pr[2].to(BF).float(), pr[3])
    yr.backward(dy.float())
    xg = x.to(DEV).requires_grad_()
    pg = [t.to(DEV).requires_grad_() for t in (w1, b1, w2, b2)]
    assert ops().mlp_gelu_ok(xg, Hd)
    yg = ops().mlp_gelu(xg, *pg)
    yg.backward(dy.to(DEV))
    assert rel(yg, yr) < 1.5e-2
    assert rel(xg.grad, xr.grad) < 2e-2
    for a, r in zip(pg, pr):
        assert rel(a.grad, r.grad) < 2e-2
    # a plain Linear on the same kernel (bias epilogue; dgrad with W^T)
    xg2 = x.to(DEV).requires_grad_()
    wg, bg = w1.to(DEV).requires_grad_(), b1.to(DEV).requires_grad_()
    assert ops().own_gemm_ok(xg2, Hd, C)
    y2 = ops().linear(xg2, wg, bg)
    d2 = rnd(M, Hd, seed=327).to(BF)
    y2.backward(d2.to(DEV))
    xr2 = x.float().requires_grad_()
    wr, br = w1.clone().requires_grad_(), b1.clone().requires_grad_()
    F_.linear(xr2, wr.to(BF).float(), br).backward(d2.float())
    assert rel(y2, F_.linear(x.float(), w1.to(BF).float(), b1)) < 1e-2
    assert rel(xg2.grad, xr2.grad) < 2e-2 and rel(wg.grad, wr.grad) < 2e-2 and rel(bg.grad, br.grad) < 2e-2


def test_grouped_weight_gradients_and_batched_fold():
    """ops.defer_folds(): the weight gradients of several Linear layers as ONE grouped launch + one batched fold must
    equal the fp32 reference of each (ragged M, N, K; with and without bias; gradients ACCUMULATED into the sinks)."""
    shapes = [(12544, 384, 384, True), (3136, 768, 768, True), (50176, 192, 96, False), (2000, 136, 104, True),
              (1500, 768, 3072, True), (200704 // 8, 96, 288, True),
              (512, 768, 3072, True), (512, 3072, 768, False), (1000, 264, 200, True), (77, 136, 72, True),   # few rows: in place
              (3136, 768, 768, True), (3648, 768, 3072, True), (3136, 2304, 768, False), (2500, 256, 512, True),   # 256 x 256 tiles
              # round 6: the former library hole — 1024 < M < 2048 (Swin stage 3 / fusion encoder at per-GPU batch 1-4) and
              # M >= 2048 with > 2^20 outputs whose widths are multiples of 64 but not of 256
              (1568, 3072, 768, True), (1696, 768, 768, True), (1824, 768, 3072, False), (1632, 2304, 768, True),
              (1025, 384, 1536, True), (2047, 1152, 384, False), (3136, 1344, 960, True), (2048, 30528 // 8, 320, True)]
    probs = []
    for i, (M, N, K, bias) in enumerate(shapes):
        dy = rnd(M, N, seed=400 + i).to(BF)
        x = rnd(M, K, seed=420 + i).to(BF)
        dw0 = rnd(N, K, seed=440 + i)
        db0 = rnd(N, seed=460 + i) if bias else None
        probs.append((dy, x, dw0, db0))
    sinks = []
    with ops().defer_folds():
        for dy, x, dw0, db0 in probs:
            dw, db = dw0.clone().to(DEV), (db0.clone().to(DEV) if db0 is not None else None)
            r = ops().linear_wgrad(dy.to(DEV), x.to(DEV), db is not None, dw, db)
            assert r == (None, None)
            sinks.append((dw, db))
        assert len(ops().WGRAD_DEFER) == len(shapes)        # every one deferred, none on the library (ops._wgrad_custom)
    assert not [k for k in ops().LIBRARY_GEMM_CALLS if k[0] == 'linear_wgrad']
    for (dy, x, dw0, db0), (dw, db) in zip(probs, sinks):
        ref = dw0 + dy.float().t() @ x.float()
        assert rel(dw, ref) < 2e-5
        if db0 is not None:
            assert rel(db, db0 + dy.float().sum(0)) < 2e-5


def test_grouped_weight_gradients_ragged_shapes_and_strided_operands():
    """The grouped launch on every Swin-T / BERT-base width pair (N, K multiples of 96): ragged M (last stage of a slice partly
    past the end), operands that are column slices of wider tensors (row stride != width), bias on / off, accumulate and
    first-touch store, in-place (few rows, very large outputs) and partial + fold paths, more problems than one launch takes.
    (Written for the shape-fitted tile class of round 5, which round 6 removed — profiles/r05_wgrad_tile_class.txt; the
    shapes stay as coverage of the fixed-tile classes.)"""
    from clover_amd import _lib
    L = _lib.lib()
    shapes = [(12545, 384, 1536, True), (3137, 1152, 384, False), (50001, 192, 192, True), (7001, 96, 96, True),
              (20011, 96, 384, True), (20000, 384, 96, False), (9000, 288, 96, True), (9999, 576, 192, True),
              (6000, 192, 768, False), (4099, 768, 192, True), (5555, 192, 384, True), (3136, 384, 768, True),
              (999, 288, 96, True), (512, 3072, 768, True), (513, 768, 2304, False), (3153, 2304, 768, True),
              (40, 480, 672, True), (2049, 960, 96, True)]
    assert all(L.clv_linear_wgrad_class(M, N, K) in (0, 1) for M, N, K, _ in shapes)
    probs = []
    for i, (M, N, K, bias) in enumerate(shapes):
        wide_y, wide_x = rnd(M, N + 24 * (i % 3), seed=900 + i).to(BF).to(DEV), rnd(M, K + 16 * (i % 2), seed=930 + i).to(BF).to(DEV)
        dy, x = wide_y[:, 8 * (i % 3):8 * (i % 3) + N], wide_x[:, :K]
        assert dy.shape == (M, N) and (i % 3 == 0 or dy.stride(0) != N)
        first = i % 4 == 1                                   # first-touch sink: stale (NaN) content, stored not added
        dw0 = torch.full((N, K), float('nan')) if first else rnd(N, K, seed=960 + i)
        db0 = rnd(N, seed=990 + i) if bias else None
        probs.append((dy, x, dw0, db0, first))
    sinks = []
    for rep in range(2):                                     # the same launches twice: nothing is left behind in the ring / tables
        sinks = []
        with ops().defer_folds():
            for dy, x, dw0, db0, first in probs:
                dw, db = dw0.clone().to(DEV), (db0.clone().to(DEV) if db0 is not None else None)
                if first:
                    st = ops().FirstTouch()
                    st.on = True
                    dw._clv_ft = st
                assert ops().linear_wgrad(dy, x, db is not None, dw, db) == (None, None)
                sinks.append((dw, db))
            assert len(ops().WGRAD_DEFER) == len(shapes)
    for (dy, x, dw0, db0, first), (dw, db), shp in zip(probs, sinks, shapes):
        ref = dy.double().t() @ x.double()
        if not first:
            ref = ref + dw0.double().to(DEV)
        assert rel(dw, ref) < 2e-5, (shp, rel(dw, ref))
        if db0 is not None:
            assert rel(db, db0.double().to(DEV) + dy.double().sum(0)) < 2e-5, shp


def test_grouped_weight_gradients_shared_sink():
    """A Linear applied twice in one backward segment: both few-row weight gradients add into the SAME dW / db in place
    (no atomics) — flush_wgrads must not put them into one launch.  Plus a many-row use of the same weight (partials)."""
    N, K = 768, 768
    dw0, db0 = rnd(N, K, seed=480), rnd(N, seed=481)
    dw, db = dw0.clone().to(DEV), db0.clone().to(DEV)
    ref_w, ref_b = dw0.clone(), db0.clone()
    with ops().defer_folds():
        for i, M in enumerate((512, 512, 640, 3136)):
            dy, x = rnd(M, N, seed=482 + i).to(BF), rnd(M, K, seed=490 + i).to(BF)
            assert ops().linear_wgrad(dy.to(DEV), x.to(DEV), True, dw, db) == (None, None)
            ref_w += dy.float().t() @ x.float()
            ref_b += dy.float().sum(0)
        assert len(ops().WGRAD_DEFER) == 4
    assert rel(dw, ref_w) < 2e-5 and rel(db, ref_b) < 2e-5


@pytest.mark.parametrize('group', ['1', '0'])           # grouped launch / per-layer partial kernels with deferred folds
def test_two_partial_mode_uses_of_one_weight(group, monkeypatch):
    """a many-row Linear applied TWICE inside one backward segment — both uses write fp32 partials and both
    folds target the same dW / db.  The batched fold adds without atomics, so the two folds must go out as separate
    launches (ops.fold_chunks); repeated to give a race the chance to show."""
    monkeypatch.setenv('CLOVER_GROUP_WGRAD', group)
    N, K = 384, 384
    for trial in range(4):
        dw0, db0 = rnd(N, K, seed=700 + trial), rnd(N, seed=710 + trial)
        dw, db = dw0.clone().to(DEV), db0.clone().to(DEV)
        ref_w, ref_b = dw0.clone().double(), db0.clone().double()
        with ops().defer_folds():
            for i, M in enumerate((3136, 3136, 12544, 12544)):
                dy, x = rnd(M, N, seed=720 + 4 * trial + i).to(BF), rnd(M, K, seed=740 + 4 * trial + i).to(BF)
                assert ops().linear_wgrad(dy.to(DEV), x.to(DEV), True, dw, db) == (None, None)
                ref_w += dy.double().t() @ x.double()
                ref_b += dy.double().sum(0)
        assert rel(dw, ref_w) < 2e-5 and rel(db, ref_b) < 2e-5, (trial, rel(dw, ref_w), rel(db, ref_b))


def test_transpose_batch():
    shapes = [(384, 1536), (96, 288), (100, 72), (64, 64), (30522 // 6, 768), (7, 5)]
    src = torch.zeros(sum(r * c for r, c in shapes) + 64, dtype=BF)
    entries, off = [], 8
    mats = []
    for i, (r, c) in enumerate(shapes):
        m = rnd(r, c, seed=310 + i).to(BF)
        src[off:off + r * c] = m.flatten()
        entries.append((off, off, r, c))
        mats.append((off, m))
        off += r * c
    sg = src.to(DEV)
    dg = torch.zeros_like(sg)
    tab, n, tiles = ops().transpose_table(entries, DEV)
    ops().transpose_batch(sg, dg, tab, n, tiles)
    out = dg.cpu()
    for (o, m) in mats:
        r, c = m.shape
        assert torch.equal(out[o:o + r * c].view(c, r), m.t().contiguous())


# ----------------------------------------------------------------------------- linear weight grad
@pytest.mark.parametrize('M,N,K', [(50176, 288, 96), (12544, 96, 384), (4096, 1152, 384), (3000, 96, 96), (777, 3072, 768),
                                   (256, 768, 3072), (1000, 200, 104), (33, 8, 8), (1500, 768, 768)])
def test_linear_and_wgrad(M, N, K):
    x = rnd(M, K, seed=71).to(BF)
    w = rnd(N, K, scale=0.05, seed=72)
    b = rnd(N, scale=0.1, seed=73)
    dy = rnd(M, N, seed=74).to(BF)
    xr, wr, br = x.float().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.linear(xr, wr.to(BF).float(), br.to(BF).float())
    yr.backward(dy.float())
    xg, wg, bg = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops().linear(xg, wg, bg)
    y.backward(dy.to(DEV))
    assert rel(y, yr) < 1e-2
    assert rel(xg.grad, xr.grad) < 1e-2
    assert rel(wg.grad, wr.grad) < 5e-3, rel(wg.grad, wr.grad)
    assert rel(bg.grad, br.grad) < 5e-3, rel(bg.grad, br.grad)


@pytest.mark.parametrize('M,N,K', [(256, 768, 768), (1024, 136, 264), (2100, 384, 96), (50176, 192, 384), (5000, 96, 96),
                                   (1568, 3072, 768), (1696, 768, 768), (1100, 200, 104), (2500, 1344, 960)])
@pytest.mark.parametrize('bias', [True, False])
def test_wgrad_accumulates_into_sink(M, N, K, bias):
    """clv_linear_wgrad with gradient sinks: dW / db are ADDED to what the fp32 slab views already hold —
    one M-slice (direct accumulation, M <= 1024) and split-M (partials + fold) alike; ragged tiles."""
    x, dy = rnd(M, K, seed=75).to(BF).to(DEV), rnd(M, N, seed=76).to(BF).to(DEV)
    dw0, db0 = rnd(N, K, seed=77).to(DEV), rnd(N, seed=78).to(DEV)
    dw, db = dw0.clone(), db0.clone()
    assert ops()._wgrad_custom(M, N, K)
    out = ops().linear_wgrad(dy, x, bias, dw, db if bias else None)
    assert out == (None, None)
    ref_w = dw0.double() + dy.double().t() @ x.double()
    assert rel(dw, ref_w.cpu()) < 1e-5, rel(dw, ref_w.cpu())
    if bias:
        assert rel(db, (db0.double() + dy.double().sum(0)).cpu()) < 1e-5
    else:
        assert torch.equal(db, db0)


def test_layernorm_fused_residual_stream():
    """y, s = LN(x + r), x + r with the gradient on s folded into the backward (Swin residual adds)."""
    rows, C = 1000, 96
    x, r = rnd(rows, C, seed=81).to(BF), rnd(rows, C, seed=82).to(BF)
    g, b = 1 + 0.1 * rnd(C, seed=83), 0.1 * rnd(C, seed=84)
    dy, ds = rnd(rows, C, seed=85).to(BF), rnd(rows, C, seed=86).to(BF)
    xr, rr = x.float().requires_grad_(), r.float().requires_grad_()
    gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
    sr = xr + rr
    yr = F.layer_norm(sr, (C,), gr, br, 1e-5)
    torch.autograd.backward([yr, sr], [dy.float(), ds.float()])
    xg, rg = x.to(DEV).requires_grad_(), r.to(DEV).requires_grad_()
    gg, bg = g.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y, s = ops().layer_norm(xg, gg, bg, 1e-5, residual=rg, return_sum=True)
    torch.autograd.backward([y, s], [dy.to(DEV), ds.to(DEV)])
    assert rel(y, yr) < 1e-2 and rel(s, sr) < 1e-2
    assert rel(xg.grad, xr.grad) < 2e-2 and rel(rg.grad, rr.grad) < 2e-2
    assert rel(gg.grad, gr.grad) < 1e-2 and rel(bg.grad, br.grad) < 1e-2


@pytest.mark.parametrize('rows,C,f32', [(1000, 96, False), (777, 192, False), (300, 384, False), (130, 768, False),
                                        (65, 1536, False), (64, 2048, False), (50, 3072, False), (40, 64, True),
                                        (37, 100, False), (9, 768, True)])
def test_layernorm_shapes(rows, C, f32):
    """Every lane-group configuration of the vector kernels (C % 8 == 0) and the scalar fallback (C = 100),
    bf16 and fp32 storage, plain and with residual + residual-stream output."""
    dt = torch.float32 if f32 else BF
    x, r = rnd(rows, C, seed=181).to(dt), rnd(rows, C, seed=182).to(dt)
    g, b = 1 + 0.1 * rnd(C, seed=183), 0.1 * rnd(C, seed=184)
    dy, ds = rnd(rows, C, seed=185).to(dt), rnd(rows, C, seed=186).to(dt)
    for with_res in (False, True):
        xr, rr = x.float().clone().requires_grad_(), r.float().clone().requires_grad_()
        gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
        sr = xr + rr if with_res else xr
        yr = F.layer_norm(sr, (C,), gr, br, 1e-5)
        xg, rg = x.to(DEV).detach().requires_grad_(), r.to(DEV).detach().requires_grad_()
        gg, bg = g.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        if with_res:
            torch.autograd.backward([yr, sr], [dy.float(), ds.float()])
            y, s_ = ops().layer_norm(xg, gg, bg, 1e-5, residual=rg, return_sum=True)
            torch.autograd.backward([y, s_], [dy.to(DEV), ds.to(DEV)])
            assert rel(rg.grad, rr.grad) < (1e-4 if f32 else 2e-2)
        else:
            yr.backward(dy.float())
            y = ops().layer_norm(xg, gg, bg, 1e-5)
            y.backward(dy.to(DEV))
        tol = 1e-4 if f32 else 2e-2
        assert rel(y, yr) < tol and rel(xg.grad, xr.grad) < tol
        assert rel(gg.grad, gr.grad) < (1e-4 if f32 else 1e-2) and rel(bg.grad, br.grad) < (1e-4 if f32 else 1e-2)


@pytest.mark.parametrize('B,T,C', [(4, 50, 96), (6, 33, 384), (2, 64, 768)])
def test_layernorm_operand_transforms(B, T, C):
    """LN(dropout(x) * xscale[b] + res): recover the kernel's dropout mask from an all-ones operand, then
    check forward, every input gradient, the residual-stream output and the forked output against torch."""
    pd = 0.2
    x, r = rnd(B, T, C, seed=191).to(BF), rnd(B, T, C, seed=192).to(BF)
    g, b = 1 + 0.1 * rnd(C, seed=193), 0.1 * rnd(C, seed=194)
    sc = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 1.25][:B])
    dy, dy2, ds = rnd(B, T, C, seed=195).to(BF), rnd(B, T, C, seed=196).to(BF), rnd(B, T, C, seed=197).to(BF)
    torch.manual_seed(5)
    o = ops()
    state = o._dropout_counter(torch.device(DEV)).clone()
    o.dropout_seeds_begin(torch.device(DEV))
    try:
        # mask probe: same seed slot (pool index 0) as the real call below, after the counter is rewound
        ones = torch.ones(B, T, C, device=DEV, dtype=BF)
        _, probe = o.layer_norm(ones, g.to(DEV), b.to(DEV), 1e-5, residual=torch.zeros_like(ones), return_sum=True,
                                x_dropout_p=pd)
        mask = (probe.float() > 0.5).float().cpu()                   # kept elements hold 1 / (1 - p)
        assert abs(mask.mean().item() - (1 - pd)) < 0.03
        o._dropout_counter(torch.device(DEV)).copy_(state)      # rewind: the next begin() redraws the same seeds
        o.dropout_seeds_begin(torch.device(DEV))
        xg, rg = x.to(DEV).requires_grad_(), r.to(DEV).requires_grad_()
        gg, bg = g.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
        y, s_, y2 = o.layer_norm(xg, gg, bg, 1e-5, residual=rg, return_sum=True, x_scale=sc.to(DEV), x_dropout_p=pd,
                                 fork=True)
        assert y2.data_ptr() == y.data_ptr()
        torch.autograd.backward([y, s_, y2], [dy.to(DEV), ds.to(DEV), dy2.to(DEV)])
    finally:
        o.dropout_seeds_end(torch.device(DEV))
    xr, rr = x.float().requires_grad_(), r.float().requires_grad_()
    gr, br = g.clone().requires_grad_(), b.clone().requires_grad_()
    sr = xr * mask / (1 - pd) * sc[:, None, None] + rr
    yr = F.layer_norm(sr, (C,), gr, br, 1e-5)
    torch.autograd.backward([yr, sr], [dy.float() + dy2.float(), ds.float()])
    assert rel(s_, sr) < 1e-2 and rel(y, yr) < 2e-2
    assert rel(xg.grad, xr.grad) < 2e-2 and rel(rg.grad, rr.grad) < 2e-2
    assert rel(gg.grad, gr.grad) < 1e-2 and rel(bg.grad, br.grad) < 1e-2
    assert float(xg.grad[0].abs().max()) == 0.0                      # sample 0 is dropped by its path factor


def test_seq_attention_dropout():
    """Attention-probability dropout: recover the kernel's mask with uniform attention + identity V, then
    check forward and backward against a torch reference that uses exactly that mask."""
    B, S, nH, hd, pd = 2, 64, 2, 64, 0.25
    Hd = nH * hd
    seed = torch.tensor([123456789], device=DEV, dtype=torch.int64)
    from clover_amd.ops import _Attention
    kw = dict(mode=0, groups=B, N=S, nH=nH, hd=hd, scale=hd ** -0.5, dropout_p=pd)
    # q = k = 0 -> P uniform 1/S ; V = identity per head -> O[i, j] = mask[i, j] / (S (1 - pd))
    probe = torch.zeros(B, S, 3, nH, hd)
    probe[:, :, 2] = torch.eye(S)[None, :, None, :hd].expand(B, S, nH, hd)
    o = _Attention.apply(probe.reshape(B, S, 3 * Hd).to(BF).to(DEV), None, None, None, kw, seed)
    mask = (o.float().view(B, S, nH, hd).permute(0, 2, 1, 3) * S * (1 - pd)).round().cpu()        # [B,nH,S(q),S(k)]
    assert set(mask.unique().tolist()) <= {0.0, 1.0}
    assert abs(mask.mean().item() - (1 - pd)) < 0.03                                         # drop rate
    assert not torch.equal(mask[0, 0], mask[1, 1])                                           # varies over (b, h)
    qkv = rnd(B, S, 3 * Hd, seed=91).to(BF)
    do = rnd(B, S, Hd, seed=92).to(BF)
    qr = qkv.float().requires_grad_()
    q, k, v = qr.view(B, S, 3, nH, hd).permute(2, 0, 3, 1, 4)
    p = (q @ k.transpose(-1, -2) / hd ** 0.5).softmax(-1) * mask / (1 - pd)
    o_ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    o_ref.backward(do.float())
    qg = qkv.to(DEV).requires_grad_()
    o2 = _Attention.apply(qg, None, None, None, kw, seed)
    o2.backward(do.to(DEV))
    assert rel(o2, o_ref) < 2e-2, rel(o2, o_ref)
    assert rel(qg.grad, qr.grad) < 3e-2, rel(qg.grad, qr.grad)


@pytest.mark.parametrize('B,S,nH,hd', [(2, 816, 12, 64), (1, 450, 2, 64), (1, 1030, 2, 32), (2, 2048, 1, 64)])
def test_long_seq_attention(B, S, nH, hd):
    """Sequences beyond 448 keys (32-frame fusion encoder: 16*49 + 32 = 816 tokens): up to 896 the fused kernels run as
    two parts of staged tokens + a merge kernel (816 and 450 here), beyond that the unfused GEMM + row-softmax path (1030,
    2048) — both against a fp32 reference; ragged lengths exercise the row / tile tails."""
    Hd = nH * hd
    qkv = rnd(B, S, 3 * Hd, seed=121).to(BF)
    do = rnd(B, S, Hd, seed=122).to(BF)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[0, S - 37:] = 0
    ext = om.extended_mask(mask)
    qr = qkv.float().requires_grad_()
    q, k, v = qr.view(B, S, 3, nH, hd).permute(2, 0, 3, 1, 4)
    p = (q @ k.transpose(-1, -2) / hd ** 0.5 + ext).softmax(-1)
    o_ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    o_ref.backward(do.float())
    qg = qkv.to(DEV).requires_grad_()
    o = ops().seq_attention(qg, ext.reshape(B, S).to(DEV).contiguous(), nH)
    o.backward(do.to(DEV))
    assert rel(o, o_ref) < 2e-2, rel(o, o_ref)
    assert rel(qg.grad, qr.grad) < 3e-2, rel(qg.grad, qr.grad)
    # masked keys receive exactly zero probability mass: their V gradient is zero
    dv = qg.grad.view(B, S, 3, nH, hd)[0, S - 37:, 2]
    assert dv.abs().max().item() == 0.0


@pytest.mark.parametrize('S', [816, 600, 896])
def test_split_seq_attention_dropout_equals_unfused_path(S):
    """The two-part fused kernels with attention-probability dropout: the mask is a hash of (seed, query row, GLOBAL key
    index), so for one seed the fused result must equal the unfused GEMM + row-softmax path's (which materialises the same
    mask) — forward and the q / k / v gradients, with a key mask."""
    from clover_amd.ops import _Attention, _LongSeqAttention
    B, nH, hd, pdrop = 2, 4, 64, 0.2
    Hd = nH * hd
    seed = torch.tensor([1234567], device=DEV, dtype=torch.int64)
    qkv = rnd(B, S, 3 * Hd, seed=141).to(BF)
    do = rnd(B, S, Hd, seed=142).to(BF).to(DEV)
    mask = torch.ones(B, S, dtype=torch.long)
    mask[1, S - 50:] = 0
    km = om.extended_mask(mask).reshape(B, S).to(DEV).contiguous()
    q1 = qkv.to(DEV).requires_grad_()
    o1 = _LongSeqAttention.apply(q1, km, nH, pdrop, seed)
    o1.backward(do)
    q2 = qkv.to(DEV).requires_grad_()
    kw = dict(mode=0, groups=B, N=S, nH=nH, hd=hd, scale=hd ** -0.5, dropout_p=pdrop)
    o2 = _Attention.apply(q2, None, None, km, kw, seed)
    o2.backward(do)
    assert rel(o2, o1) < 2e-2, rel(o2, o1)
    assert rel(q2.grad, q1.grad) < 3e-2, rel(q2.grad, q1.grad)


def test_long_seq_attention_dropout():
    """Dropout on the long path: recover the mask with uniform attention + identity V, then forward / backward
    against a torch reference using exactly that mask; the mask is the fused kernels' (same hash of seed, row, key)."""
    from clover_amd.ops import _Attention, _LongSeqAttention
    B, S, nH, hd, pdrop = 1, 512, 2, 64, 0.25
    Hd = nH * hd
    seed = torch.tensor([987654321], device=DEV, dtype=torch.int64)
    masks = []
    for blk in range(S // hd):                      # V = one 64-column slice of the identity per pass
        probe = torch.zeros(B, S, 3, nH, hd)
        probe[:, blk * hd:(blk + 1) * hd, 2] = torch.eye(hd)[None, :, None, :].expand(B, hd, nH, hd)
        o = _LongSeqAttention.apply(probe.reshape(B, S, 3 * Hd).to(BF).to(DEV), None, nH, pdrop, seed)
        masks.append((o.float().view(B, S, nH, hd).permute(0, 2, 1, 3) * S * (1 - pdrop)).round().cpu())
    mask = torch.cat(masks, dim=-1)                                                            # [B,nH,S(q),S(k)]
    assert set(mask.unique().tolist()) <= {0.0, 1.0}
    assert abs(mask.mean().item() - (1 - pdrop)) < 0.01
    # the fused kernel draws the same mask for the rows / keys both can address (S = 64 sub-problem has its own
    # row ids, so compare through the hash contract instead: group 0, head 0 rows are ids 0..S-1 in both)
    kw = dict(mode=0, groups=1, N=64, nH=1, hd=64, scale=64 ** -0.5, dropout_p=pdrop)
    probe = torch.zeros(1, 64, 3, 1, 64)
    probe[:, :, 2] = torch.eye(64)[None, :, None, :]
    of = _Attention.apply(probe.reshape(1, 64, 192).to(BF).to(DEV), None, None, None, kw, seed)
    mf = (of.float().view(64, 64) * 64 * (1 - pdrop)).round().cpu()
    assert torch.equal(mf, mask[0, 0, :64, :64])
    qkv = rnd(B, S, 3 * Hd, seed=131).to(BF)
    do = rnd(B, S, Hd, seed=132).to(BF)
    qr = qkv.float().requires_grad_()
    q, k, v = qr.view(B, S, 3, nH, hd).permute(2, 0, 3, 1, 4)
    p = (q @ k.transpose(-1, -2) / hd ** 0.5).softmax(-1) * mask / (1 - pdrop)
    o_ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    o_ref.backward(do.float())
    qg = qkv.to(DEV).requires_grad_()
    o2 = _LongSeqAttention.apply(qg, None, nH, pdrop, seed)
    o2.backward(do.to(DEV))
    assert rel(o2, o_ref) < 2e-2, rel(o2, o_ref)
    assert rel(qg.grad, qr.grad) < 3e-2, rel(qg.grad, qr.grad)


# ----------------------------------------------------------------------------- row-streaming GEMM
@pytest.mark.parametrize('M,N,K', [(5000, 288, 96), (3001, 96, 384), (2000, 768, 192), (1000, 192, 768), (777, 96, 288)])
def test_rowgemm_plain_and_gelu_bwd(M, N, K):
    x = rnd(M, K, seed=101).to(BF)
    w = rnd(N, K, scale=0.1, seed=102).to(BF)
    b = rnd(N, scale=0.1, seed=103)
    ref = x.float() @ w.float().t() + b
    out = ops().rowgemm(x.to(DEV), w.to(DEV), b.to(DEV))
    assert rel(out['y'], ref) < 1e-2
    pre = rnd(M, N, seed=104).to(BF)
    xr = pre.float().requires_grad_()
    om.gelu(xr).backward(ref - b)                                  # = (x W^T) * gelu'(pre)
    out2 = ops().rowgemm(x.to(DEV), w.to(DEV), None, epilogue=2, pre_in=pre.to(DEV))
    assert rel(out2['y'], xr.grad) < 2e-2


@pytest.mark.parametrize('M,N,K,with_res,gelu', [(5000, 288, 96, True, False), (3001, 384, 96, True, True),
                                                 (2000, 576, 192, False, False), (1000, 768, 192, True, True),
                                                 (900, 384, 128, False, True)])
def test_rowgemm_layernorm_prologue(M, N, K, with_res, gelu):
    x = rnd(M, K, seed=111).to(BF)
    r = rnd(M, K, seed=112).to(BF) if with_res else None
    w = rnd(N, K, scale=0.1, seed=113).to(BF)
    b = rnd(N, scale=0.1, seed=114)
    s = x.float() + (r.float() if with_res else 0)
    xh = F.layer_norm(s, (K,), None, None, 1e-5)
    pre = xh @ w.float().t() + b
    out = ops().rowgemm(x.to(DEV), w.to(DEV), b.to(DEV), res=r.to(DEV) if with_res else None, standardise=True,
                        epilogue=1 if gelu else 0)
    assert rel(out['y'], om.gelu(pre) if gelu else pre) < 1.5e-2
    if gelu:
        assert rel(out['pre'], pre) < 1.5e-2
    if with_res:
        assert rel(out['sum'], s) < 1e-2
    assert rel(out['mean'], s.mean(-1)) < 1e-3 and rel(out['rstd'], (s.var(-1, unbiased=False) + 1e-5).rsqrt()) < 1e-3


@pytest.mark.parametrize('C,with_res', [(96, True), (128, False)])
def test_fused_ln_linear_and_mlp(C, with_res):
    """Fused residual + LayerNorm + projection, and the fused MLP block, against the plain fp32 composition."""
    M, Hd = 3000, 4 * C
    a, r = rnd(M, C, seed=121).to(BF), rnd(M, C, seed=122).to(BF)
    P = dict(g=1 + 0.1 * rnd(C, seed=123), b=0.1 * rnd(C, seed=124), wq=rnd(3 * C, C, scale=0.1, seed=125),
             bq=0.1 * rnd(3 * C, seed=126), w1=rnd(Hd, C, scale=0.1, seed=127), b1=0.1 * rnd(Hd, seed=128),
             w2=rnd(C, Hd, scale=0.05, seed=129), b2=0.1 * rnd(C, seed=130))
    dq, dm, ds = rnd(M, 3 * C, seed=131).to(BF), rnd(M, C, seed=132).to(BF), rnd(M, C, seed=133).to(BF)

    def run(dev, fused):
        ar, rr = a.to(dev).float().requires_grad_(), r.to(dev).float().requires_grad_()
        Q = {k: v.to(dev).clone().requires_grad_() for k, v in P.items()}
        if fused:
            ab, rb = ar.to(BF), (rr.to(BF) if with_res else None)
            q, s = ops().ln_linear(ab, rb, Q['g'], Q['b'], Q['wq'], Q['bq'])
            m, s2 = ops().fused_mlp(ab, rb, Q['g'], Q['b'], Q['w1'], Q['b1'], Q['w2'], Q['b2'])
            outs, grads = [q, m], [dq.to(dev), dm.to(dev)]
            if with_res:
                outs += [s, s2]
                grads += [ds.to(dev), ds.to(dev)]
        else:
            s = ar + rr if with_res else ar
            y = F.layer_norm(s, (C,), Q['g'], Q['b'], 1e-5)
            q = F.linear(y, Q['wq'], Q['bq'])
            m = F.linear(om.gelu(F.linear(y, Q['w1'], Q['b1'])), Q['w2'], Q['b2'])
            outs, grads = [q, m], [dq.float(), dm.float()]
            if with_res:
                outs += [s, s]
                grads += [ds.float(), ds.float()]
        torch.autograd.backward(outs, grads)
        return outs, ar.grad, (rr.grad if with_res else None), {k: v.grad for k, v in Q.items()}

    o_ref, ga_ref, gr_ref, gp_ref = run('cpu', False)
    o, ga, gr, gp = run(DEV, True)
    assert rel(o[0], o_ref[0]) < 2e-2 and rel(o[1], o_ref[1]) < 2e-2
    assert rel(ga, ga_ref) < 3e-2
    if with_res:
        assert rel(gr, gr_ref) < 3e-2 and rel(o[2], o_ref[2]) < 1e-2
    for k in P:
        assert rel(gp[k], gp_ref[k]) < 3e-2, (k, rel(gp[k], gp_ref[k]))


@pytest.mark.parametrize('C', [96, 128])
def test_fused_mlp_with_drop_path_factor(C):
    """fused_mlp(x_scale=...): the per-sample DropPath factor of the attention branch inside the residual add of the
    LayerNorm + fc1 kernel (t = factor[b] * a + r) and inside its backward (d a = factor[b] * d t, d r = d t) — against the
    plain fp32 composition; a dropped sample (factor 0) gets no gradient into its branch."""
    B, rows_per, Hd = 4, 700, 4 * C
    a, r = rnd(B, rows_per, C, seed=141).to(BF), rnd(B, rows_per, C, seed=142).to(BF)
    sc = torch.tensor([1.25, 0.0, 1.25, 1.25])
    P = dict(g=1 + 0.1 * rnd(C, seed=143), b=0.1 * rnd(C, seed=144), w1=rnd(Hd, C, scale=0.1, seed=145),
             b1=0.1 * rnd(Hd, seed=146), w2=rnd(C, Hd, scale=0.05, seed=147), b2=0.1 * rnd(C, seed=148))
    dm, ds = rnd(B, rows_per, C, seed=149).to(BF), rnd(B, rows_per, C, seed=150).to(BF)
    ar, rr = a.float().requires_grad_(), r.float().requires_grad_()
    Q = {k: v.clone().requires_grad_() for k, v in P.items()}
    t = ar * sc.view(B, 1, 1) + rr
    y = F.layer_norm(t, (C,), Q['g'], Q['b'], 1e-5)
    m_ref = F.linear(om.gelu(F.linear(y, Q['w1'], Q['b1'])), Q['w2'], Q['b2'])
    torch.autograd.backward([m_ref, t], [dm.float(), ds.float()])
    ag, rg = a.to(DEV).requires_grad_(), r.to(DEV).requires_grad_()
    G = {k: v.to(DEV).clone().requires_grad_() for k, v in P.items()}
    m, s2 = ops().fused_mlp(ag, rg, G['g'], G['b'], G['w1'], G['b1'], G['w2'], G['b2'], x_scale=sc.to(DEV))
    torch.autograd.backward([m, s2], [dm.to(DEV), ds.to(DEV)])
    assert rel(m, m_ref) < 2e-2 and rel(s2, t) < 1e-2
    assert rel(ag.grad, ar.grad) < 3e-2 and rel(rg.grad, rr.grad) < 3e-2
    assert float(ag.grad[1].abs().max()) == 0.0
    for k in P:
        assert rel(G[k].grad, Q[k].grad) < 3e-2, (k, rel(G[k].grad, Q[k].grad))


@pytest.mark.parametrize('M', [16, 17, 33, 1000, 3001, 25088, 200704 // 4])
@pytest.mark.parametrize('with_res,with_scale', [(True, True), (True, False), (False, False)])
def test_mlp_one_kernel_forward_backward(M, with_res, with_scale):
    """clv_mlp_fused_fwd / _bwd (norm2 + fc1 + GELU + fc2 of a VideoSwin-T stage-0 block in ONE kernel each way, both
    weight matrices in LDS, hidden activations never written: swin_transformer_3d.py:482-483,262-268,503) through the raw
    launchers: every output against an fp32 torch composition on the same bf16 operands — out, the residual stream, the
    statistics; d a, d r, and the act / d pre / xhat operands the backward leaves for the weight-gradient kernels.  Ragged
    row counts (a half group, a partial tile, fewer tiles than workgroups), with / without residual and DropPath factor."""
    F_ = torch.nn.functional
    C, Hd, rps = 96, 384, max(1, M // 4)
    nb = (M + rps - 1) // rps
    a, r = rnd(M, C, seed=151).to(BF), rnd(M, C, seed=152).to(BF)
    w1f, b1f = rnd(Hd, C, scale=0.12, seed=153).to(BF), 0.1 * rnd(Hd, seed=154)
    w2, b2 = rnd(C, Hd, scale=0.06, seed=155).to(BF), 0.1 * rnd(C, seed=156)
    sc = (torch.tensor([1.25, 0.0, 1.25, 1.25, 1.25])[:nb] if with_scale else None)
    do, ds = rnd(M, C, seed=157).to(BF), rnd(M, C, seed=158).to(BF)
    # fp32 reference on the bf16-rounded quantities the kernels see
    ar, rr = a.float().requires_grad_(), r.float().requires_grad_()
    t = ar * (sc.repeat_interleave(rps)[:M, None] if with_scale else 1.0) + (rr if with_res else 0.0)
    mu = t.mean(1, keepdim=True)
    var = ((t - mu) ** 2).mean(1, keepdim=True)
    xhat = (t - mu) * torch.rsqrt(var + 1e-5)
    pre = F_.linear(xhat, w1f.float(), b1f)
    act = om.gelu(pre)
    out = F_.linear(act, w2.float(), b2)
    out.backward(do.float(), retain_graph=True)
    da_mlp, dr_mlp = ar.grad.clone(), (rr.grad.clone() if with_res else None)
    L = ops()
    o = L.mlp_fused_fwd(a.to(DEV), r.to(DEV) if with_res else None, w1f.to(DEV), b1f.to(DEV), w2.to(DEV), b2.to(DEV), 1e-5,
                        sc.to(DEV) if with_scale else None, rps)
    assert rel(o['out'], out.detach()) < 1.5e-2, rel(o['out'], out.detach())
    assert rel(o['mean'], mu.detach().squeeze(1)) < 1e-4 and rel(o['rstd'], torch.rsqrt(var + 1e-5).detach().squeeze(1)) < 1e-4
    if with_res:
        assert rel(o['sum'], t.detach()) < 5e-3
    ts = o['sum'] if with_res else a.to(DEV)
    bw = L.mlp_fused_bwd(ts, o['mean'], o['rstd'], do.to(DEV), ds.to(DEV), w1f.to(DEV), b1f.to(DEV),
                         w2.t().contiguous().to(DEV), sc.to(DEV) if with_scale else None, rps, want_dres=with_scale)
    assert rel(bw['act'], act.detach()) < 1.5e-2 and rel(bw['xhat'], xhat.detach()) < 1e-2
    dact = do.float() @ w2.float()
    x = pre.detach().requires_grad_()
    om.gelu(x).backward(torch.ones_like(x))
    assert rel(bw['dpre'], dact * x.grad) < 1.5e-2, rel(bw['dpre'], dact * x.grad)
    # d t = LN-backward(d pre W1f) + d sum;  d a = factor * d t,  d r = d t
    fac = sc.repeat_interleave(rps)[:M, None] if with_scale else torch.ones(M, 1)
    dt_mlp = dr_mlp if with_res else da_mlp                               # = LN-backward part (factor 1 when no residual)
    dt = dt_mlp + ds.float()
    assert rel(bw['da'], dt * fac) < 2e-2, rel(bw['da'], dt * fac)
    if with_scale:
        assert rel(bw['dres'], dt) < 2e-2
        if nb > 1 and rps < M:
            assert float(bw['da'][rps:min(2 * rps, M)].abs().max()) == 0.0         # the dropped sample: no gradient into a
    torch.cuda.synchronize()


def test_mlp_one_kernel_equals_two_kernel_path(monkeypatch):
    """ops.fused_mlp at C = 96 with CLOVER_FUSED_MLP=1 (one kernel each way) against CLOVER_FUSED_MLP=0 (the round-5 path:
    LayerNorm + fc1 + GELU kernel, fc2 GEMM, three-kernel backward) on identical operands: outputs and every gradient."""
    B, rows_per, C, Hd = 4, 1568, 96, 384
    a, r = rnd(B, rows_per, C, seed=161).to(BF), rnd(B, rows_per, C, seed=162).to(BF)
    sc = torch.tensor([1.25, 0.0, 1.25, 1.25])
    P = dict(g=1 + 0.1 * rnd(C, seed=163), b=0.1 * rnd(C, seed=164), w1=rnd(Hd, C, scale=0.1, seed=165),
             b1=0.1 * rnd(Hd, seed=166), w2=rnd(C, Hd, scale=0.05, seed=167), b2=0.1 * rnd(C, seed=168))
    dm, ds = rnd(B, rows_per, C, seed=169).to(BF), rnd(B, rows_per, C, seed=170).to(BF)
    res = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('CLOVER_FUSED_MLP', mode)
        ag, rg = a.to(DEV).requires_grad_(), r.to(DEV).requires_grad_()
        G = {k: v.to(DEV).clone().requires_grad_() for k, v in P.items()}
        m, s2 = ops().fused_mlp(ag, rg, G['g'], G['b'], G['w1'], G['b1'], G['w2'], G['b2'], x_scale=sc.to(DEV))
        torch.autograd.backward([m, s2], [dm.to(DEV), ds.to(DEV)])
        res[mode] = dict(m=m.detach(), s=s2.detach(), da=ag.grad, dr=rg.grad, **{k: v.grad for k, v in G.items()})
    assert torch.equal(res['1']['s'], res['0']['s'])
    for k in res['1']:
        assert rel(res['1'][k], res['0'][k].float().cpu()) < 1.5e-2, (k, rel(res['1'][k], res['0'][k].float().cpu()))


def test_gemm_nt_gelu_grad_epilogues():
    """CLV_GEMM_EPI_BIAS_GELU_D (c2 = GELU'(pre)) and CLV_GEMM_EPI_MUL (c = acc * aux): the pair that keeps the GELU
    backward free of transcendentals, against torch fp32."""
    M, N, K = 12544, 1536, 384
    x, w, b = rnd(M, K, seed=501).to(BF), rnd(N, K, scale=0.05, seed=502).to(BF), rnd(N, seed=503)
    act, dg = ops().gemm_nt(x.to(DEV), w.to(DEV), b.to(DEV), epilogue=ops().GEMM_EPI_BIAS_GELU_D)
    pre = (x.float() @ w.float().t() + b.to(BF).float()).requires_grad_()
    ref = F.gelu(pre)
    ref.sum().backward()
    assert rel(act, ref.detach()) < 1e-2
    assert (dg.float().cpu() - pre.grad).abs().max().item() < 1.5e-2          # bf16 storage of values in [-0.13, 1.13]
    dy = rnd(M, K, seed=504).to(BF)
    wt = rnd(N, K, scale=0.05, seed=505).to(BF)
    out = ops().gemm_nt(dy.to(DEV), wt.to(DEV), aux=dg, epilogue=ops().GEMM_EPI_MUL)
    ref2 = (dy.float() @ wt.float().t()) * dg.float().cpu()
    assert rel(out, ref2) < 1e-2


# ----------------------------------------------------------------------------- MLM decoder on the own kernels
@pytest.mark.parametrize('R,V,H', [(256, 30522, 768), (64, 1003, 128), (40, 2050, 256)])
def test_mlm_decoder_padded_vocabulary(R, V, H):
    """ops.mlm_decoder + the strided focal loss (BertLMPredictionHead.decoder + SoftmaxFocalLossMultiClass,
    mlm_itm_head.py:38-41, focal_loss.py:61-72) with the vocabulary padded to a multiple of 64 by phantom rows, as the
    engine lays the parameters out: scores, loss, input gradient, weight / bias gradients against fp32 torch; the phantom
    rows of the gradients stay exactly zero."""
    F_ = torch.nn.functional
    L = ops()
    Vp = (V + 63) // 64 * 64
    x = rnd(R, H, seed=601).to(BF)
    w = rnd(V, H, scale=0.05, seed=602)
    b = rnd(V, scale=0.1, seed=603)
    labels = torch.full((R,), -100, dtype=torch.long)
    sel = torch.arange(0, R, 3)
    labels[sel] = torch.randint(0, V, (len(sel),), generator=torch.Generator().manual_seed(604))
    # reference (bf16-rounded operands, fp32 math)
    xr = x.float().requires_grad_()
    wr = w.to(BF).float().requires_grad_()
    br = b.to(BF).float().requires_grad_()
    logits_r = xr @ wr.t() + br
    ce = F_.cross_entropy(logits_r[sel], labels[sel], reduction='none')
    loss_r = (((1 - torch.exp(-ce)) ** 2) * ce).mean()
    loss_r.backward()
    # engine-style padded views
    weight = torch.nn.Parameter(w.to(DEV))
    bias = torch.nn.Parameter(b.to(DEV))
    wp = torch.zeros(Vp, H, device=DEV)
    wp[:V] = w.to(DEV)
    bp = torch.zeros(Vp, device=DEV)
    bp[:V] = b.to(DEV)
    weight._clv_pad_shadow = wp.to(BF)
    weight._clv_pad_shadow_t = wp.to(BF).t().contiguous()
    weight._clv_pad_grad = torch.zeros(Vp, H, device=DEV)
    bias._clv_pad_weight = bp
    bias._clv_pad_grad = torch.zeros(Vp, device=DEV)
    weight._clv_ready = bias._clv_ready = lambda: None
    xg = x.to(DEV).requires_grad_()
    assert L.mlm_decoder_ok(xg, weight, bias)
    scores = L.mlm_decoder(xg, weight, bias)
    assert scores.shape == (R, V) and scores.stride() == (Vp, 1)
    assert rel(scores, logits_r.detach()) < 1e-2
    loss = L.focal_ce_masked(scores, labels.to(DEV), 2.0)
    assert abs(loss.item() - loss_r.item()) < 2e-3 * max(1.0, abs(loss_r.item()))
    before = dict(L.MLM_DECODER_STATS)
    loss.backward()
    # the focal backward's own padded gradient buffer is contracted over in place (it carries the marker), not copied
    assert L.MLM_DECODER_STATS['in_place'] == before['in_place'] + 1 and L.MLM_DECODER_STATS['copied'] == before['copied']
    assert rel(xg.grad, xr.grad) < 3e-2, rel(xg.grad, xr.grad)
    assert rel(weight._clv_pad_grad[:V], wr.grad) < 3e-2
    assert rel(bias._clv_pad_grad[:V], br.grad) < 3e-2
    if Vp > V:
        assert float(weight._clv_pad_grad[V:].abs().max()) == 0.0 and float(bias._clv_pad_grad[V:].abs().max()) == 0.0
    # a gradient that is a [R, Vp]-strided view of some OTHER live buffer (here: of the scores buffer itself) is
    # copied, never written — its padding columns keep their content
    weight._clv_pad_grad.zero_(); bias._clv_pad_grad.zero_(); xg.grad = None
    scores2 = L.mlm_decoder(xg, weight, bias)
    if Vp > V:
        pad_before = scores2._base[:, V:].clone()
        before = dict(L.MLM_DECODER_STATS)
        scores2.backward(scores2.detach())                  # d scores = a view of the scores buffer
        assert L.MLM_DECODER_STATS['copied'] == before['copied'] + 1
        assert torch.equal(scores2._base[:, V:], pad_before)
    # round 6: WITHOUT engine views (plain parameters) the same kernels run on per-call padded operands and autograd
    # receives [V, H] / [V] gradients — no library GEMM
    L.LIBRARY_GEMM_CALLS.clear()
    w2, b2 = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    xg2 = x.to(DEV).requires_grad_()
    assert L.mlm_decoder_ok(xg2, w2, b2) == (H % 64 == 0)
    if H % 64 == 0:
        sc = L.mlm_decoder(xg2, w2, b2)
        assert sc.shape == (R, V) and rel(sc, logits_r.detach()) < 1e-2
        with L.defer_folds():                               # even inside a deferring segment the gradients arrive at once
            L.focal_ce_masked(sc, labels.to(DEV), 2.0).backward()
        assert rel(xg2.grad, xr.grad) < 3e-2 and rel(w2.grad, wr.grad) < 3e-2 and rel(b2.grad, br.grad) < 3e-2
        assert w2.grad.shape == (V, H) and b2.grad.shape == (V,)
        assert not L.LIBRARY_GEMM_CALLS, L.LIBRARY_GEMM_CALLS


# ----------------------------------------------------------------------------- BatchNorm variants of the projection heads
@pytest.mark.parametrize('B,D', [(8, 768), (16, 1536), (5, 96), (300, 1000)])
def test_batchnorm1d_kernel(B, D):
    """clv_batchnorm1d_fwd / _bwd against torch.nn.BatchNorm1d (fp32, CPU): training mode (batch statistics, running
    averages with the unbiased variance, the gradient through the statistics) and eval mode (running statistics)."""
    from clover_amd.nn import BatchNorm1d
    x = rnd(B, D, seed=1300) * 2.0 + 0.5
    dy = rnd(B, D, seed=1301)
    ref, own = torch.nn.BatchNorm1d(D), BatchNorm1d(D).to(DEV)
    with torch.no_grad():
        ref.weight.copy_(1 + 0.1 * rnd(D, seed=1302)); ref.bias.copy_(0.1 * rnd(D, seed=1303))
        ref.running_mean.copy_(0.2 * rnd(D, seed=1304)); ref.running_var.copy_(1 + 0.3 * rnd(D, seed=1305).abs())
    own.load_state_dict(ref.state_dict())
    assert set(own.state_dict()) == set(ref.state_dict())
    for mode in ('train', 'train', 'eval'):
        ref.train(mode == 'train'); own.train(mode == 'train')
        xr, xg = x.clone().requires_grad_(), x.to(DEV).requires_grad_()
        ref.zero_grad(); own.zero_grad()
        yr, y = ref(xr), own(xg)
        yr.backward(dy); y.backward(dy.to(DEV))
        assert rel(y, yr) < 2e-5, (mode, rel(y, yr))
        assert rel(xg.grad, xr.grad) < 2e-4, (mode, rel(xg.grad, xr.grad))
        assert rel(own.weight.grad, ref.weight.grad) < 2e-5 and rel(own.bias.grad, ref.bias.grad) < 2e-5
        assert rel(own.running_mean, ref.running_mean) < 1e-5 and rel(own.running_var, ref.running_var) < 1e-5
        assert int(own.num_batches_tracked) == int(ref.num_batches_tracked)


def test_heads_batchnorm_variants():
    """NCEHeadForMM(ln=False, text_bn=True), NCEHeadForVision(ln=False), NCEHeadForText(text_bn=True)
    (ssl_head.py:50-66,175-186,252-262): same state_dict keys as the reference's module tree, outputs and parameter gradients
    equal to an fp32 torch restatement of that tree; a batch that stacks two forward passes (the recognizer's doubled clean +
    masked pass) is normalised per pass, in the reference's call order."""
    import torch.nn as nn
    from clover_amd.heads.ssl_head import NCEHeadForMM, NCEHeadForText, NCEHeadForVision
    torch.manual_seed(7)
    Cv, Ct, Hd, E, B = 96, 64, 128, 64, 6
    mm = NCEHeadForMM(Cv, Ct, Hd, E, ln=False, text_bn=True, dropout_ratio=0, text_agg_type='cls').to(DEV).train()
    ref_img = nn.Sequential(nn.Linear(Cv, Hd), nn.BatchNorm1d(Hd), nn.GELU(), nn.Linear(Hd, E), nn.BatchNorm1d(E))
    ref_txt = nn.Sequential(nn.Linear(Ct, Ct), nn.BatchNorm1d(Ct), nn.GELU(), nn.Linear(Ct, E))
    assert set(mm.img_projector.state_dict()) == set(ref_img.state_dict())
    assert set(mm.text_projector.state_dict()) == set(ref_txt.state_dict())
    ref_img.load_state_dict({k: v.cpu() for k, v in mm.img_projector.state_dict().items()})
    ref_txt.load_state_dict({k: v.cpu() for k, v in mm.text_projector.state_dict().items()})
    vis = rnd(2 * B, 2, 3, 3, Cv, seed=1310)                  # channels-last [2B, T, h, w, C]: clean ; masked
    txt = rnd(2 * B, 5, Ct, seed=1311)                        # [masked captions ; un-masked captions]
    w = rnd(2 * B, E, seed=1312)
    y = mm.forward_vision(vis.to(DEV), channels_last=True, passes=2)
    pooled = vis.mean(dim=(1, 2, 3))
    yr = torch.cat([ref_img(pooled[:B]), ref_img(pooled[B:])])          # clean first (:102), masked second (:159)
    assert rel(y, yr) < 1e-4, rel(y, yr)
    t = mm.forward_text(txt.to(DEV), passes=2, order=(1, 0))
    cls = txt[:, 0]
    t1 = ref_txt(cls[B:])                                               # un-masked first (:102) ...
    tr = torch.cat([ref_txt(cls[:B]), t1])                              # ... masked second (:150)
    assert rel(t, tr) < 1e-4, rel(t, tr)
    ((y * w.to(DEV)).sum() + (t * w.to(DEV)).sum()).backward()
    ((yr * w).sum() + (tr * w).sum()).backward()
    for own_m, ref_m in ((mm.img_projector, ref_img), (mm.text_projector, ref_txt)):
        for (n, p), (_, q) in zip(own_m.named_parameters(), ref_m.named_parameters()):
            # (the bias of a Linear that feeds a BatchNorm has an exactly-zero gradient — the batch mean is removed — so both
            # sides hold 1e-6 round-off there: absolute floor)
            err = float((p.grad.cpu() - q.grad).abs().max())
            assert err <= 5e-4 * float(q.grad.abs().max()) + 2e-5, (n, err)
        for (n, b_), (_, c_) in zip(own_m.named_buffers(), ref_m.named_buffers()):
            assert rel(b_.float(), c_.float()) < 1e-5, n                # running statistics after two calls, in order
    v = NCEHeadForVision(cross_in_channels=Ct, visual_in_channels=Ct, hidden_dim=32, vts_embed_dim=E, ln=False,
                         dropout_ratio=0).to(DEV).train()
    assert isinstance(v.img_bn1, nn.BatchNorm1d) and isinstance(v.img_bn2, nn.BatchNorm1d)
    tt = NCEHeadForText(cross_in_channels=Ct, vts_embed_dim=E, text_bn=True, dropout_ratio=0).to(DEV).train()
    assert isinstance(tt.bn, nn.BatchNorm1d)
    xin = rnd(B, Ct, seed=1313)
    rv = nn.Sequential(nn.Linear(Ct, 64), nn.BatchNorm1d(64), nn.GELU(), nn.Linear(64, E), nn.BatchNorm1d(E))
    rv.load_state_dict({k2: v.state_dict()[k1].cpu() for k1, k2 in zip(
        [f'{m}.{s}' for m in ('img_fc1', 'img_bn1', 'img_fc2', 'img_bn2') for s in
         (('weight', 'bias') if 'fc' in m else ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'))],
        [f'{i}.{s}' for i, m in ((0, 'fc'), (1, 'bn'), (3, 'fc'), (4, 'bn')) for s in
         (('weight', 'bias') if m == 'fc' else ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked'))])})
    assert rel(v(xin.to(DEV)), rv(xin)) < 1e-4
    rt = nn.Sequential(nn.Linear(Ct, Ct), nn.BatchNorm1d(Ct), nn.GELU(), nn.Linear(Ct, E))
    rt.load_state_dict({k2: tt.state_dict()[k1].cpu() for k1, k2 in zip(
        ['fc1.weight', 'fc1.bias', 'bn.weight', 'bn.bias', 'bn.running_mean', 'bn.running_var', 'bn.num_batches_tracked',
         'fc2.weight', 'fc2.bias'],
        ['0.weight', '0.bias', '1.weight', '1.bias', '1.running_mean', '1.running_var', '1.num_batches_tracked',
         '3.weight', '3.bias'])})
    assert rel(tt(xin.to(DEV)), rt(xin)) < 1e-4
    # under an ablation switch the reference skips one of the doubled passes (masked captions without
    # mlm_ssl_V_head :147-150, masked clips without symmetry_rank :155-159).  That block is still computed here (its slot of
    # the packed embeddings exists) but must not reach the BatchNorm layers' running statistics: `live` names the blocks
    # the reference runs; buffers afterwards equal those of a torch reference that saw ONLY the live block.
    for own_m, ref_m in ((mm.img_projector, ref_img), (mm.text_projector, ref_txt)):
        ref_m.load_state_dict({k: v_.cpu() for k, v_ in own_m.state_dict().items()})
    with torch.no_grad():
        y2 = mm.forward_vision(vis.to(DEV), channels_last=True, passes=2, live=(0,))
        t2 = mm.forward_text(txt.to(DEV), passes=2, order=(1, 0), live=(1,))
        yr2, tr2 = ref_img(pooled[:B]), ref_txt(cls[B:])                 # the reference runs only these
    assert rel(y2[:B], yr2) < 1e-4 and rel(t2[B:], tr2) < 1e-4
    assert y2.shape[0] == 2 * B and bool(torch.isfinite(y2).all()) and bool(torch.isfinite(t2).all())
    for own_m, ref_m in ((mm.img_projector, ref_img), (mm.text_projector, ref_txt)):
        for (n, b_), (_, c_) in zip(own_m.named_buffers(), ref_m.named_buffers()):
            assert rel(b_.float(), c_.float()) < 1e-5, n                # one momentum update / one tracked batch, not two
