"""ORACLE (test infrastructure only) — fp32 CPU restatement of Clover's
video-text pre-training step (forward + losses; backward through torch autograd).

Plain functional PyTorch over a flat ``{name: tensor}`` parameter dict that uses
the reference's own ``state_dict`` key names.  Floating-point path, so the
restatement is in torch fp32 (the tier's "keep a torch fp32 reference only for a
floating-point kernel" clause); the integer/index logic lives in
``oracle/indexing.py`` (numpy) and is what this file calls for every index.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; the product (``clover_amd``) never does.

PINNING: every function here is checked against outputs of the reference itself
(imported in the build container by ``tests/golden/ref_harness.py``); fixtures
and the generating script are under ``tests/golden/``.  The BERT / fusion-encoder
/ MLM-transform arithmetic lives in the un-vendored ``transformers==4.6.1``
(install.sh:25); the reference holds no tests, so those parts are pinned by the
same goldens (generated with the installed transformers 5.x BERT modules patched
to 4.6.1 mask semantics) — see DESIGN.md "Oracle".

All citations are relative to /root/reference/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import indexing as ix


# --------------------------------------------------------------------------- #
# small helpers
# --------------------------------------------------------------------------- #
def linear(P, name, x):
    b = P.get(name + '.bias')
    return F.linear(x, P[name + '.weight'], b)


def layer_norm(P, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), P[name + '.weight'], P[name + '.bias'], eps)


def gelu(x):
    return F.gelu(x)  # exact erf GELU (nn.GELU default; HF 'gelu')


# --------------------------------------------------------------------------- #
# video encoder: SwinTransformer3D
# --------------------------------------------------------------------------- #
SWIN_DEFAULTS = dict(patch_size=(2, 4, 4), in_chans=3, embed_dim=96, depths=[2, 2, 6, 2],
                     num_heads=[3, 6, 12, 24], window_size=(8, 7, 7), mlp_ratio=4.,
                     patch_norm=True)


def swin_cfg(cfg):
    c = dict(SWIN_DEFAULTS)
    c.update({k: v for k, v in cfg.items() if k in c})
    return c


def patch_embed(P, pre, x, cfg):
    """PatchEmbed3D.forward — swin_transformer_3d.py:671-688.
    x [B,3,D,H,W] -> [B,C,D',H',W'] (zero right/bottom/back pad, conv3d k=s, LN(C))."""
    pd, ph, pw = cfg['patch_size']
    _, _, D, H, W = x.shape
    if W % pw != 0:
        x = F.pad(x, (0, pw - W % pw))
    if H % ph != 0:
        x = F.pad(x, (0, 0, 0, ph - H % ph))
    if D % pd != 0:
        x = F.pad(x, (0, 0, 0, 0, 0, pd - D % pd))
    x = F.conv3d(x, P[pre + 'proj.weight'], P[pre + 'proj.bias'], stride=cfg['patch_size'])
    if cfg['patch_norm']:
        Dd, Wh, Ww = x.shape[2:]
        x = x.flatten(2).transpose(1, 2)
        x = layer_norm(P, pre + 'norm', x, 1e-5)
        x = x.transpose(1, 2).reshape(-1, cfg['embed_dim'], Dd, Wh, Ww)
    return x


def window_attention(P, pre, xw, mask, num_heads, rel_index_full):
    """WindowAttention3D.forward — swin_transformer_3d.py:369-400.
    xw [B_,N,C]; mask [nW,N,N] or None; rel_index_full = index of the CONFIGURED window."""
    B_, N, C = xw.shape
    hd = C // num_heads
    scale = hd ** -0.5
    qkv = linear(P, pre + 'qkv', xw).reshape(B_, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    q = q * scale
    attn = q @ k.transpose(-2, -1)
    idx = torch.from_numpy(rel_index_full[:N, :N].reshape(-1).copy())
    bias = P[pre + 'relative_position_bias_table'][idx].reshape(N, N, -1).permute(2, 0, 1).contiguous()
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = attn.view(B_ // nW, nW, num_heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
        attn = attn.view(-1, num_heads, N, N)
    attn = attn.softmax(dim=-1)
    x = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return linear(P, pre + 'proj', x)


def _t_window_partition(x, ws):
    B, D, H, W, C = x.shape
    x = x.view(B, D // ws[0], ws[0], H // ws[1], ws[1], W // ws[2], ws[2], C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous().view(-1, ws[0] * ws[1] * ws[2], C)


def _t_window_reverse(w, ws, B, D, H, W):
    x = w.view(B, D // ws[0], H // ws[1], W // ws[2], ws[0], ws[1], ws[2], -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).contiguous().view(B, D, H, W, -1)


def swin_block(P, pre, x, cfg_ws, block_shift, num_heads, mask_matrix, rel_index_full):
    """SwinTransformerBlock3D.forward (drop_path = identity) —
    swin_transformer_3d.py:446-505.  x [B,D,H,W,C]."""
    B, D, H, W, C = x.shape
    ws, ss = ix.get_window_size((D, H, W), cfg_ws, block_shift)
    shortcut = x
    x = layer_norm(P, pre + 'norm1', x, 1e-5)
    pad_d1 = (ws[0] - D % ws[0]) % ws[0]
    pad_b = (ws[1] - H % ws[1]) % ws[1]
    pad_r = (ws[2] - W % ws[2]) % ws[2]
    x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b, 0, pad_d1))
    _, Dp, Hp, Wp, _ = x.shape
    if any(i > 0 for i in ss):
        shifted = torch.roll(x, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
        attn_mask = mask_matrix
    else:
        shifted = x
        attn_mask = None
    xw = _t_window_partition(shifted, ws)
    aw = window_attention(P, pre + 'attn.', xw, attn_mask, num_heads, rel_index_full)
    aw = aw.view(-1, *(ws + (C,)))
    shifted = _t_window_reverse(aw, ws, B, Dp, Hp, Wp)
    if any(i > 0 for i in ss):
        x = torch.roll(shifted, shifts=(ss[0], ss[1], ss[2]), dims=(1, 2, 3))
    else:
        x = shifted
    if pad_d1 > 0 or pad_r > 0 or pad_b > 0:
        x = x[:, :D, :H, :W, :].contiguous()
    x = shortcut + x
    # forward_part2: mlp(norm2(x))  (:482-483, Mlp :262-268)
    h = layer_norm(P, pre + 'norm2', x, 1e-5)
    h = linear(P, pre + 'mlp.fc2', gelu(linear(P, pre + 'mlp.fc1', h)))
    return x + h


def patch_merging(P, pre, x):
    """PatchMerging.forward — swin_transformer_3d.py:521-544."""
    B, D, H, W, C = x.shape
    if (H % 2 == 1) or (W % 2 == 1):
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x0 = x[:, :, 0::2, 0::2, :]
    x1 = x[:, :, 1::2, 0::2, :]
    x2 = x[:, :, 0::2, 1::2, :]
    x3 = x[:, :, 1::2, 1::2, :]
    x = torch.cat([x0, x1, x2, x3], -1)
    x = layer_norm(P, pre + 'norm', x, 1e-5)
    return F.linear(x, P[pre + 'reduction.weight'])


def swin_forward(P, pre, x, cfg, mask=None, taps=None):
    """SwinTransformer3D.forward — swin_transformer_3d.py:218-242.
    x [B,3,T,H,W]; mask [B,1,mh,mw] int or None.  Returns [B,Cf,T',h,w] (and w)."""
    cfg = swin_cfg(cfg)
    x = patch_embed(P, pre + 'patch_embed.', x, cfg)
    if taps is not None:
        taps['patch_embed'] = x
    w = None
    if mask is not None:
        B, Dc, T, H, W = x.shape
        w_np = ix.mask_blend_weight(mask.cpu().numpy(), T, H, W)
        w = torch.from_numpy(w_np).to(x.dtype)
        mask_tokens = P[pre + 'mask_token'].expand(B, -1, T, H, W)
        x = x * (1. - w) + mask_tokens * w
    rel_index_full = ix.relative_position_index(cfg['window_size'])
    n_layers = len(cfg['depths'])
    for i in range(n_layers):
        # BasicLayer.forward — :625-646
        B, C, D, H, W = x.shape
        cfg_ws = tuple(cfg['window_size'])
        cfg_ss = tuple(s // 2 for s in cfg_ws)
        ws, ss = ix.get_window_size((D, H, W), cfg_ws, cfg_ss)
        x = x.permute(0, 2, 3, 4, 1)
        Dp = int(np.ceil(D / ws[0])) * ws[0]
        Hp = int(np.ceil(H / ws[1])) * ws[1]
        Wp = int(np.ceil(W / ws[2])) * ws[2]
        attn_mask = torch.from_numpy(ix.compute_mask(Dp, Hp, Wp, ws, ss))
        for j in range(cfg['depths'][i]):
            blk_shift = (0, 0, 0) if j % 2 == 0 else cfg_ss
            x = swin_block(P, f'{pre}layers.{i}.blocks.{j}.', x, cfg_ws, blk_shift,
                           cfg['num_heads'][i], attn_mask, rel_index_full)
            if taps is not None:
                taps[f'layers.{i}.blocks.{j}'] = x
        x = x.reshape(B, D, H, W, -1)
        if i < n_layers - 1:
            x = patch_merging(P, f'{pre}layers.{i}.downsample.', x)
        x = x.permute(0, 4, 1, 2, 3)
    x = x.permute(0, 2, 3, 4, 1)
    x = layer_norm(P, pre + 'norm', x, 1e-5)
    x = x.permute(0, 4, 1, 2, 3)
    if mask is not None:
        return x, w
    return x


# --------------------------------------------------------------------------- #
# BERT (transformers 4.6.1 semantics; call sites bert_from_hugface.py:30,
# cross_transformer.py:109-110, mlm_itm_head.py:33-41)
# --------------------------------------------------------------------------- #
def bert_layer(P, pre, h, ext_mask, heads, eps):
    """One HF BertLayer (post-LN): self-attn -> dense+res+LN -> FFN(gelu) -> dense+res+LN."""
    B, S, Hd = h.shape
    hd = Hd // heads

    def split(t):
        return t.view(B, S, heads, hd).permute(0, 2, 1, 3)
    q = split(linear(P, pre + 'attention.self.query', h))
    k = split(linear(P, pre + 'attention.self.key', h))
    v = split(linear(P, pre + 'attention.self.value', h))
    scores = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if ext_mask is not None:
        scores = scores + ext_mask
    probs = scores.softmax(dim=-1)
    ctx = (probs @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
    a = layer_norm(P, pre + 'attention.output.LayerNorm',
                   linear(P, pre + 'attention.output.dense', ctx) + h, eps)
    inter = gelu(linear(P, pre + 'intermediate.dense', a))
    return layer_norm(P, pre + 'output.LayerNorm', linear(P, pre + 'output.dense', inter) + a, eps)


def extended_mask(mask):
    """transformers 4.6.1 get_extended_attention_mask for a 2-D mask:
    (1 - mask[:,None,None,:]) * -10000.0  (cross_transformer.py:109)."""
    return (1.0 - mask[:, None, None, :].to(torch.float32)) * -10000.0


def bert_encoder(P, pre, h, ext_mask, n_layers, heads, eps):
    for i in range(n_layers):
        h = bert_layer(P, f'{pre}layer.{i}.', h, ext_mask, heads, eps)
    return h


def bert_embeddings(P, pre, ids, eps):
    """HF BertEmbeddings: word + token_type(0) + absolute position, LN."""
    L = ids.shape[1]
    e = P[pre + 'word_embeddings.weight'][ids]
    e = e + P[pre + 'token_type_embeddings.weight'][torch.zeros_like(ids)]
    e = e + P[pre + 'position_embeddings.weight'][torch.arange(L)][None]
    return layer_norm(P, pre + 'LayerNorm', e, eps)


def bert_forward(P, pre, ids, mask, bcfg):
    """BertFromPretrained.forward -> last_hidden_state — bert_from_hugface.py:26-32.
    bcfg: dict(num_hidden_layers, num_attention_heads, layer_norm_eps)."""
    h = bert_embeddings(P, pre + 'bert.embeddings.', ids, bcfg['layer_norm_eps'])
    return bert_encoder(P, pre + 'bert.encoder.', h, extended_mask(mask), bcfg['num_hidden_layers'],
                        bcfg['num_attention_heads'], bcfg['layer_norm_eps'])


# --------------------------------------------------------------------------- #
# fusion transformer
# --------------------------------------------------------------------------- #
def fusion_forward(P, pre, visual_token, text_mask, text_embeds, fcfg):
    """CrossModalTransformerFromPretrained.forward (use_text_cls=True, no prompt) —
    cross_transformer.py:64-124.  visual_token [B,T,S,Din]; returns dict."""
    if (pre + 'fc_in.weight') in P:
        visual_token = linear(P, pre + 'fc_in', visual_token)       # :69-70
    B, T, S, D = visual_token.shape
    tt = P[pre + 'token_type_embeddings.weight']
    text = text_embeds + tt[1]                                       # :84-86
    v = visual_token + P[pre + 'vis_space_pos'] + P[pre + 'vis_tempor_pos'][:, :T]   # :89
    v = v.contiguous().view(B, T * S, D)
    v = v + tt[0]                                                    # :92-94
    v = layer_norm(P, pre + 'norm', v, 1e-5)                         # :97 (nn.LayerNorm default eps)
    feat = torch.cat([v, text], dim=1)                               # :108
    mm_mask = torch.cat([torch.ones(B, T * S, dtype=text_mask.dtype), text_mask], dim=1)
    out = bert_encoder(P, pre + 'bert_encoder.', feat, extended_mask(mm_mask),
                       fcfg['num_hidden_layers'], fcfg['num_attention_heads'], fcfg['layer_norm_eps'])
    return {'last_hidden_state': out,
            't_last_hidden_state': out[:, T * S:],                   # :117
            'v_last_hidden_state': out[:, :T * S]}                   # :118


# --------------------------------------------------------------------------- #
# heads
# --------------------------------------------------------------------------- #
def nce_mm_forward_vision(P, pre, img):
    """NCEHeadForMM.forward_vision (ln=True, dropout 0) — ssl_head.py:103-116."""
    x = img.mean(dim=(2, 3, 4))                                      # AdaptiveAvgPool3d(1)
    x = linear(P, pre + 'img_projector.0', x)
    x = layer_norm(P, pre + 'img_projector.1', x, 1e-5)
    x = gelu(x)
    x = linear(P, pre + 'img_projector.3', x)
    return layer_norm(P, pre + 'img_projector.4', x, 1e-5)


def nce_mm_forward_text(P, pre, text):
    """NCEHeadForMM.forward_text (text_agg_type='cls', text_bn=False) — ssl_head.py:118-139."""
    x = text[:, 0]
    return linear(P, pre + 'text_projector.2', gelu(linear(P, pre + 'text_projector.0', x)))


def nce_vision_head(P, pre, x):
    """NCEHeadForVision.forward on a 2-D CLS row (defect R1: mean(dim=1) treated as
    identity, i.e. input fed as [B,1,D]) — ssl_head.py:200-221."""
    x = linear(P, pre + 'img_fc1', x)
    x = layer_norm(P, pre + 'img_bn1', x, 1e-5)
    x = gelu(x)
    x = linear(P, pre + 'img_fc2', x)
    return layer_norm(P, pre + 'img_bn2', x, 1e-5)


def nce_text_head(P, pre, x):
    """NCEHeadForText.forward (dropout in eval) — ssl_head.py:275-297."""
    return linear(P, pre + 'fc2', gelu(linear(P, pre + 'fc1', x)))


def mlm_head(P, pre, h):
    """MLMHead -> BertLMPredictionHead — mlm_itm_head.py:38-52 (HF transform+decoder)."""
    h = linear(P, pre + 'predictions.transform.dense', h)
    h = gelu(h)
    h = layer_norm(P, pre + 'predictions.transform.LayerNorm', h, 1e-12)
    return linear(P, pre + 'predictions.decoder', h)


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
def focal_loss_multiclass(logits, target, gamma=2.0):
    """SoftmaxFocalLossMultiClass.forward — focal_loss.py:61-72."""
    ce = F.cross_entropy(logits, target, reduction='none')
    pt = torch.exp(-ce)
    return ((1 - pt) ** gamma * ce).mean()


def cos_norm(a, eps=1e-8):
    """contrastive_loss.py:20-25."""
    n = a.norm(dim=-1)[:, None]
    return a / torch.max(n, eps * torch.ones_like(n))


class _VariedShapeGather(torch.autograd.Function):
    """VariedShapeGatherLoss — gather_loss.py:24-72 (rank-major concat; backward =
    the local slice only, no reduction: reference behaviour R6)."""

    @staticmethod
    def forward(ctx, q, rank, ws):
        import torch.distributed as dist
        ctx.rank = rank
        local = torch.tensor(q.size(0))
        sizes = [torch.zeros_like(local) for _ in range(ws)]
        dist.all_gather(sizes, local)
        mx = max(int(s) for s in sizes)
        ctx.cum = torch.tensor([int(s) for s in sizes]).cumsum(0).tolist()
        if mx - q.size(0):
            q = torch.cat((q, torch.zeros((mx - q.size(0),) + q.shape[1:], dtype=q.dtype)))
        outs = [torch.zeros_like(q) for _ in range(ws)]
        dist.all_gather(outs, q.contiguous())
        return torch.cat([o[:int(s)] for o, s in zip(outs, sizes)], dim=0)

    @staticmethod
    def backward(ctx, g):
        s = ctx.cum[ctx.rank - 1] if ctx.rank > 0 else 0
        return g[s:ctx.cum[ctx.rank]], None, None


def gather_rows(x):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return _VariedShapeGather.apply(x, dist.get_rank(), dist.get_world_size())
    return x


def exclusive_nce_rank_loss(video, text, text_mask, text_recon, temperature=0.05, margin=5.0,
                            use_rank=True, gather=True):
    """ExclusiveNCEwithRankingLoss.forward — contrastive_loss.py:103-161."""
    if gather:
        video, text, text_mask, text_recon = (gather_rows(t) for t in (video, text, text_mask, text_recon))
    v, t, tm, tr = cos_norm(video), cos_norm(text), cos_norm(text_mask), cos_norm(text_recon)
    sim_vt = v @ t.t() / temperature
    sim_vtm = v @ tm.t() / temperature
    sim_vtr = v @ tr.t() / temperature
    vt_d, vtm_d, vtr_d = torch.diag(sim_vt), torch.diag(sim_vtm), torch.diag(sim_vtr)
    ex = lambda s, d: s - torch.diag_embed(d + 10000.)               # noqa: E731  (:130-132)
    f_vt = torch.cat([sim_vt, ex(sim_vtm, vtm_d), ex(sim_vtr, vtr_d)], dim=1)
    f_vtm = torch.cat([ex(sim_vt, vt_d), sim_vtm, ex(sim_vtr, vtr_d)], dim=1)
    f_vtr = torch.cat([ex(sim_vt, vt_d), ex(sim_vtm, vtm_d), sim_vtr], dim=1)
    Bn = f_vt.size(0)
    a = F.log_softmax(f_vt, dim=1)[:, :Bn]
    b = F.log_softmax(f_vtm, dim=1)[:, Bn:2 * Bn]
    c = F.log_softmax(f_vtr, dim=1)[:, 2 * Bn:3 * Bn]
    diag_all = torch.diag(a) + torch.diag(b) + torch.diag(c)
    loss_v = -(diag_all.sum() / len(diag_all))
    t2v = torch.cat([sim_vt, sim_vtm, sim_vtr], dim=1).t()
    lsm = F.log_softmax(t2v, dim=1).view(-1, t2v.shape[1], t2v.shape[1])
    loss_t = -torch.mean(lsm.diagonal(dim1=1, dim2=2).mean(dim=1))
    losses = {'nce_loss': loss_v + loss_t}
    if use_rank:
        # MarginRankingLoss(margin)(x1, x2, y=1) = mean(max(0, -(x1-x2) + margin))   (:155-159)
        losses['rank_t_tm_loss'] = torch.clamp(-(vt_d - vtm_d) + margin, min=0).mean()
    return losses


# --------------------------------------------------------------------------- #
# the step graph
# --------------------------------------------------------------------------- #
def forward_train(P, batch, cfg, gather=True):
    """CloverPretrain.forward_train — multimodal_transformer_pretrain.py:76-173.

    cfg: dict(backbone=<swin kwargs>, bert=<dict(num_hidden_layers, num_attention_heads,
    layer_norm_eps)>, fusion=<same keys>, temperature, margin, gamma, vocab).
    batch: imgs [B,1,3,T,H,W], token_ids/input_mask/mlm_label [B,1,L], v_token_mask [B,1,mh,mw].
    """
    imgs = batch['imgs'].reshape((-1,) + batch['imgs'].shape[2:])
    token_ids = batch['token_ids'].reshape((-1,) + batch['token_ids'].shape[2:])
    tmask = batch['input_mask'].reshape((-1,) + batch['input_mask'].shape[2:])
    mlm_label = batch['mlm_label'].reshape((-1,) + batch['mlm_label'].shape[2:])
    v_token_mask = batch['v_token_mask']

    visual_token = swin_forward(P, 'backbone.', imgs, cfg['backbone'])            # :91
    B, D, T, H, W = visual_token.shape
    ssl_ids = torch.where(mlm_label == -100, token_ids, mlm_label)                 # :97
    text_no_mask = bert_forward(P, 'text_backbone.', ssl_ids, tmask, cfg['bert'])  # :99-101
    visual_emb = nce_mm_forward_vision(P, 'ssl_head.', visual_token)               # :102
    text_emb = nce_mm_forward_text(P, 'ssl_head.', text_no_mask)
    vt = visual_token.reshape(B, D, T, -1).permute(0, 2, 3, 1)                     # :106
    text_with_mask = bert_forward(P, 'text_backbone.', token_ids, tmask, cfg['bert'])   # :110-111
    visual_masked, _w = swin_forward(P, 'backbone.', imgs.clone(), cfg['backbone'], v_token_mask)  # :114
    vtm = visual_masked.reshape(B, D, T, -1).permute(0, 2, 3, 1)
    v_fusion = fusion_forward(P, 'multimodal_backbone.', vtm, tmask, text_no_mask, cfg['fusion'])     # :117
    t_fusion = fusion_forward(P, 'multimodal_backbone.', vt, tmask, text_with_mask, cfg['fusion'])    # :119
    t_last = t_fusion['t_last_hidden_state']

    # ablation switches of the recognizer's constructor (:26-43), all on in the published recipe:
    # cfg['mlm_head'] (:129), cfg['mlm_ssl_head'] (:147: the V / T reconstruction heads), cfg['symmetry_rank'] (:155)
    losses = {}
    if cfg.get('mlm_head', True):
        scores = mlm_head(P, 'mlm_head.', t_last)                                  # :134
        rows = torch.where(mlm_label.reshape(-1) != -100)[0]                       # :137
        losses['mlm_loss'] = focal_loss_multiclass(scores.reshape(-1, scores.shape[-1])[rows],
                                                   mlm_label.reshape(-1)[rows], cfg.get('gamma', 2.0))

    kw = dict(temperature=cfg.get('temperature', 0.05), margin=cfg.get('margin', 5.0), gather=gather)
    if cfg.get('mlm_ssl_head', True):                                              # :147
        mask_visual_recon = nce_vision_head(P, 'mlm_ssl_V_head.', v_fusion['t_last_hidden_state'][:, 0])  # :148-149
        mask_word_emb = nce_mm_forward_text(P, 'ssl_head.', text_with_mask)        # :150
        losses.update(exclusive_nce_rank_loss(visual_emb, text_emb, mask_word_emb, mask_visual_recon, **kw))  # :151

    if cfg.get('symmetry_rank', True):                                             # :155
        mask_word_recon = nce_text_head(P, 'mlm_ssl_T_head.', t_last[:, 0])        # :156-157
        mask_visual_emb = nce_mm_forward_vision(P, 'ssl_head.', visual_masked)     # :159
        l2 = exclusive_nce_rank_loss(text_emb, visual_emb, mask_visual_emb, mask_word_recon, **kw)   # :161
        losses['v_nce_loss'] = l2['nce_loss']                                      # :162
        losses['rank_v_vm_loss'] = l2['rank_t_tm_loss']                            # :165
    return losses


# --------------------------------------------------------------------------- #
# retrieval fine-tuning
# --------------------------------------------------------------------------- #
def norm_softmax_loss(video=None, text=None, sim_mat=None, temperature=0.07, cos_sim=False, gather=True):
    """NormSoftmaxLoss.forward — contrastive_loss.py:39-68."""
    if sim_mat is None:
        if gather:
            video, text = gather_rows(video), gather_rows(text)
        if cos_sim:                                                   # sim_matrix :10-18, eps 1e-8
            x = cos_norm(video) @ cos_norm(text).t() / temperature
        else:                                                         # F.normalize :51-53, eps 1e-12
            vn = video / video.norm(dim=-1, keepdim=True).clamp_min(1e-12)
            tn = text / text.norm(dim=-1, keepdim=True).clamp_min(1e-12)
            x = vn @ tn.t() / temperature
    else:
        x = sim_mat
    idiag = torch.diag(F.log_softmax(x, dim=1))
    jdiag = torch.diag(F.log_softmax(x.t(), dim=1))
    return -(idiag.sum() / len(idiag)) - (jdiag.sum() / len(jdiag))


def recall_metrics(video, text):
    """recall_for_video_text_retrieval — core/evaluation/accuracy.py:430-462 (numpy; rows with zero norm
    are left unscaled, mmaction/utils/numpy_norm.py:5-8).  Returns [R@1, R@5, R@10, MR, Recall@all]."""
    def unit(a):
        n = np.linalg.norm(a, axis=-1)
        n[n == 0] = 1
        return a / n[:, None]
    scores = unit(np.asarray(text)) @ unit(np.asarray(video)).T
    rank = np.where(np.argsort(-scores, axis=1) == np.arange(len(scores))[:, None])[1]
    r = [100.0 * np.mean(rank == 0), 100.0 * np.mean(rank < 5), 100.0 * np.mean(rank < 10), np.median(rank) + 1]
    return np.array(r + [r[0] + r[1] + r[2] - r[3]])


def finetune_embeddings(P, batch, cfg):
    """CloverFinetune retrieval encoders — multimodal_transformer_finetune.py:61-84 (train) and :128-148
    (separate_test): (visual_emb, text_emb).  imgs [B, clips, 3, T, H, W]; several clips per sample are
    averaged after the video encoder (:73-75)."""
    imgs = batch['imgs'].reshape((-1,) + batch['imgs'].shape[2:])
    B_text = batch['token_ids'].shape[0]
    token_ids = batch['token_ids'].reshape((-1,) + batch['token_ids'].shape[2:])
    tmask = batch['input_mask'].reshape((-1,) + batch['input_mask'].shape[2:])
    visual_token = swin_forward(P, 'backbone.', imgs, cfg['backbone'])
    if B_text != visual_token.shape[0]:
        visual_token = visual_token.reshape((B_text, -1) + visual_token.shape[1:]).mean(dim=1)
    text = bert_forward(P, 'text_backbone.', token_ids, tmask, cfg['bert'])
    return nce_mm_forward_vision(P, 'ssl_head.', visual_token), nce_mm_forward_text(P, 'ssl_head.', text)


def finetune_forward_train(P, batch, cfg, temperature=0.05, cos_sim=True, gather=True):
    """CloverFinetune.forward_train(task='retrieval') — multimodal_transformer_finetune.py:59-86."""
    v, t = finetune_embeddings(P, batch, cfg)
    return {'retrieval_nce_loss': norm_softmax_loss(v, t, temperature=temperature, cos_sim=cos_sim, gather=gather)}


def parse_losses(losses):
    """BaseRecognizer._parse_losses — recognizers/base.py:254-288 (single process:
    loss = sum of every key containing 'loss'; log_vars as python floats)."""
    import torch.distributed as dist
    log_vars = {k: v.mean() for k, v in losses.items()}
    loss = sum(v for k, v in log_vars.items() if 'loss' in k)
    log_vars['loss'] = loss
    out = {}
    for k, v in log_vars.items():
        v = v.detach().clone()
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(v.div_(dist.get_world_size()))
        out[k] = v.item()
    return loss, out
