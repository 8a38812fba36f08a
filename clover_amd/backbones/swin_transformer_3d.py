"""Video Swin Transformer backbone, MI355X-native.

Registered as ``SwinTransformer3D`` with the reference's constructor kwargs and
``state_dict`` key names (mmaction/models/backbones/swin_transformer_3d.py:18-247), so
reference / VideoSwin checkpoints load unchanged.  What differs is how it runs:

* activations stay channels-last ``[B,T',H,W,C]`` bf16 end-to-end (the reference bounces
  between ``B C D H W`` and ``B D H W C`` at every stage, :634,645,237-239);
* patch embedding + LayerNorm + mask-token blend is one HIP kernel that reads the fp32 clip
  once and emits the clean AND the masked token tensors (``clv_patch_embed_fwd``);
* cyclic shift, window partition, relative-position bias, shift mask, softmax, PV, window
  reverse and un-shift are one HIP kernel on the natural token layout (``clv_attn_fwd``):
  no rolled / partitioned copies and no [B_,nH,N,N] score tensor ever reach HBM;
* ``forward_pair`` runs the clean and the masked pass of the pre-training step
  (multimodal_transformer_pretrain.py:91,114) as ONE pass over 2B clips — every weight is
  read once and every GEMM has twice the rows; results equal two separate passes because
  no op mixes samples (LayerNorm only).
"""
from functools import lru_cache

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as _checkpoint

from .. import ops
from ..builder import BACKBONES
from ..nn import GELU, DropPath, LayerNorm, Linear, trunc_normal_

BF16 = ops.BF16          # the 16-bit storage type: bf16, or fp16 with CLOVER_HALF=f16 (the name is historical)


# --------------------------------------------------------------------------- geometry (host, cached)
def get_window_size(x_size, window_size, shift_size=None):
    """Clamp window to the feature size, zero the shift on clamped axes (reference :302-315)."""
    use_ws = list(window_size)
    use_ss = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            use_ws[i] = x_size[i]
            if use_ss is not None:
                use_ss[i] = 0
    if shift_size is None:
        return tuple(use_ws)
    return tuple(use_ws), tuple(use_ss)


def build_relative_position_index(window_size):
    """[N,N] int64 index into the (2wd-1)(2wh-1)(2ww-1)-row bias table (reference :345-359)."""
    wd, wh, ww = window_size
    coords = torch.stack(torch.meshgrid(torch.arange(wd), torch.arange(wh), torch.arange(ww), indexing='ij'))
    cf = torch.flatten(coords, 1)
    rel = (cf[:, :, None] - cf[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += wd - 1
    rel[:, :, 1] += wh - 1
    rel[:, :, 2] += ww - 1
    rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
    rel[:, :, 1] *= (2 * ww - 1)
    return rel.sum(-1)


def region_id_windows(Dp, Hp, Wp, window_size, shift_size):
    """Region ids of compute_mask (reference :548-562) per (window, token): int32 [nW, N].
    The reference's additive mask is ``-100 * (rid[w,i] != rid[w,j])``; the attention kernel
    evaluates that predicate in registers instead of reading an [nW,N,N] tensor."""
    img = np.zeros((Dp, Hp, Wp), dtype=np.int32)
    cnt = 0
    ws, ss = window_size, shift_size
    # same python slice triples as the reference, degenerate cases (shift 0) included
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[d, h, w] = cnt
                cnt += 1
    x = img.reshape(Dp // ws[0], ws[0], Hp // ws[1], ws[1], Wp // ws[2], ws[2])
    return np.ascontiguousarray(x.transpose(0, 2, 4, 1, 3, 5).reshape(-1, ws[0] * ws[1] * ws[2]))


@lru_cache(maxsize=64)
def _window_geometry_cached(x_size, cfg_ws, cfg_ss, device_str):
    ws, ss = get_window_size(x_size, cfg_ws, cfg_ss)
    padded = tuple(int(np.ceil(x_size[i] / ws[i])) * ws[i] for i in range(3))
    rid = None
    if any(s > 0 for s in ss):
        rid = torch.from_numpy(region_id_windows(padded[0], padded[1], padded[2], ws, ss)).to(device_str)
    return ws, ss, rid, padded


def window_geometry(x_size, cfg_ws, cfg_ss, device):
    """(effective window, effective shift, region ids on `device` or None)."""
    ws, ss, rid, _ = _window_geometry_cached(tuple(x_size), tuple(cfg_ws), tuple(cfg_ss), str(device))
    return ws, ss, rid


def gathered_bias(table, rel_index, N):
    """table[index[:N,:N]] -> fp32 [nH, N, N] (reference :382-384).  Not on the hot path any more — the attention
    kernels read the table itself (LDS) — kept as the host-side statement of what they compute."""
    nH = table.shape[1]
    return table.float()[rel_index[:N, :N].reshape(-1)].view(N, N, nH).permute(2, 0, 1).contiguous()


def mask_blend_weight(mask, T, H, W):
    """w [B,1,T,H,W] of the reference's mask-token blend (:226-229)."""
    _, _, mh, mw = mask.shape
    w = mask.unsqueeze(-1).unsqueeze(-3).expand(-1, T, -1, H // mh, -1, W // mw)
    return w.flatten(2, 3).flatten(3, 4).unsqueeze(1)


# --------------------------------------------------------------------------- modules
class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        for lin in (self.fc1, self.fc2):                 # engine keeps W^T where the input gradient's kernel takes it
            lin.weight._clv_want_t = ops.wants_transposed(lin.out_features, lin.in_features)
        if ops.mlp_fused_shape(in_features, hidden_features) and out_features == in_features:
            self.fc2.weight._clv_want_t = True           # the one-kernel MLP backward contracts d out with W2^T [hidden][C]

    def forward(self, x):
        if (type(self.act) is GELU and self.drop.p == 0.0 and self.fc1.bias is not None
                and ops.mlp_gelu_ok(x, self.fc1.out_features)):
            return ops.mlp_gelu(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class WindowAttention3D(nn.Module):
    """Parameters of the reference module (:331-367); the compute is ``ops.window_attention``
    on the un-partitioned token grid."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.dim = dim
        self.window_size = window_size
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        if qk_scale is not None and abs(qk_scale - head_dim ** -0.5) > 1e-12:
            raise NotImplementedError('qk_scale override is not supported by the HIP attention kernel')
        if attn_drop != 0.:
            raise NotImplementedError('attn_drop_rate > 0 is not supported (Clover configs use 0)')
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size[0] - 1) * (2 * window_size[1] - 1) * (2 * window_size[2] - 1), num_heads))
        self.register_buffer('relative_position_index', build_relative_position_index(window_size))
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = Linear(dim, dim)
        for lin in (self.qkv, self.proj):                # engine keeps W^T where the input gradient's kernel takes it
            lin.weight._clv_want_t = ops.wants_transposed(lin.out_features, lin.in_features)
        self.proj_drop = nn.Dropout(proj_drop)
        trunc_normal_(self.relative_position_bias_table, std=.02)

    def forward(self, x, ws, ss, rid):
        """x bf16 [B,Dp,Hp,Wp,C] (already LN'd and padded) -> same shape."""
        N = ws[0] * ws[1] * ws[2]
        qkv = self.qkv(x)
        o = ops.window_attention(qkv, self.relative_position_bias_table, rid if any(s > 0 for s in ss) else None,
                                 ws, ss, self.num_heads, table_window=self.window_size)
        return self.proj_drop(self.proj(o))


class SwinTransformerBlock3D(nn.Module):
    def __init__(self, dim, num_heads, window_size=(2, 7, 7), shift_size=(0, 0, 0), mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=GELU, norm_layer=LayerNorm,
                 use_checkpoint=False):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.window_size, self.shift_size = window_size, shift_size
        self.mlp_ratio = mlp_ratio
        self.use_checkpoint = use_checkpoint
        for i in range(3):
            assert 0 <= shift_size[i] < window_size[i], 'shift_size must in 0-window_size'
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention3D(dim, window_size=window_size, num_heads=num_heads, qkv_bias=qkv_bias,
                                      qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def attn_part(self, x):
        """Everything of forward_part1 after norm1 (x is already normalised)."""
        B, D, H, W, C = x.shape
        ws, ss, rid = window_geometry((D, H, W), self.window_size, self.shift_size, x.device)
        pad_d1 = (ws[0] - D % ws[0]) % ws[0]
        pad_b = (ws[1] - H % ws[1]) % ws[1]
        pad_r = (ws[2] - W % ws[2]) % ws[2]
        if pad_d1 or pad_b or pad_r:
            x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b, 0, pad_d1))
        x = self.attn(x, ws, ss, rid)
        if pad_d1 or pad_b or pad_r:
            x = x[:, :D, :H, :W, :].contiguous()
        return x

    def forward_part1(self, x):
        return self.attn_part(self.norm1(x))

    def forward_part2(self, x):
        return self.drop_path(self.mlp(self.norm2(x)))

    def _dp_scale(self, x):
        return self.drop_path.scale(x) if isinstance(self.drop_path, DropPath) else None

    def forward_pending(self, x, branch=None, bscale=None):
        """Residual stream with a PENDING addition: the block input is x (+ bscale * branch); returns
        (stream, branch', bscale') with block output = stream + bscale' * branch'.  Both residual adds of the
        reference (:498, :503) AND the DropPath factors on the branches are folded into the LayerNorm kernels
        that follow them (fwd: x_scale / sum_out, bwd: dsum)."""
        if self._fused_ok(x):
            return self._forward_pending_fused(x, DropPath.apply_scale(branch, bscale) if branch is not None else None)
        if branch is None:
            y1, s0 = self.norm1(x, return_sum=True)          # s0 = x (a view): its gradient folds into norm1's backward
        else:
            y1, s0 = self.norm1(branch, residual=x, return_sum=True, x_scale=bscale)
        a = self.attn_part(y1)
        y2, s1 = self.norm2(a, residual=s0, return_sum=True, x_scale=self._dp_scale(a))
        m = self.mlp(y2)
        return s1, m, self._dp_scale(m)

    def _fused_ok(self, x):
        B, D, H, W, C = x.shape
        ws = get_window_size((D, H, W), self.window_size)
        no_pad = D % ws[0] == 0 and H % ws[1] == 0 and W % ws[2] == 0
        return (x.is_cuda and no_pad and self.mlp.drop.p == 0.0 and self.attn.proj_drop.p == 0.0
                and self.attn.qkv.bias is not None and type(self.mlp.act) is GELU
                and ops.fused_block_supported(C, self.mlp.fc1.out_features))

    def _forward_pending_fused(self, x, branch):
        """Stage-0 widths (C <= 128, HBM-bound GEMMs): residual add + LayerNorm + projection fused into the
        row-streaming MFMA kernel (qkv; fc1 with GELU), GELU backward fused into the fc2 dgrad."""
        B, D, H, W, C = x.shape
        ws, ss, rid = window_geometry((D, H, W), self.window_size, self.shift_size, x.device)
        at = self.attn
        if branch is None:
            qkv, s0 = ops.ln_linear(x, None, self.norm1.weight, self.norm1.bias, at.qkv.weight, at.qkv.bias,
                                    self.norm1.eps, stream_out=True)
        else:
            qkv, s0 = ops.ln_linear(branch, x, self.norm1.weight, self.norm1.bias, at.qkv.weight, at.qkv.bias,
                                    self.norm1.eps)
        o = ops.window_attention(qkv, at.relative_position_bias_table, rid if any(s > 0 for s in ss) else None, ws, ss,
                                 self.num_heads, table_window=at.window_size)
        a = at.proj(o)
        # the branch's DropPath factor rides in the residual add of the LayerNorm + fc1 kernel (and in its backward)
        m, s1 = ops.fused_mlp(a, s0, self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias,
                              self.mlp.fc2.weight, self.mlp.fc2.bias, self.norm2.eps, x_scale=self._dp_scale(a))
        return s1, m, self._dp_scale(m)        # the consumer applies / folds the factor (next LayerNorm or merge)

    def forward(self, x, mask_matrix=None):
        """x bf16 [B,D,H,W,C].  ``mask_matrix`` is accepted for signature compatibility and unused:
        the shift mask is evaluated from region ids inside the kernel."""
        s, m, sc = self.forward_pending(x)
        return s + DropPath.apply_scale(m, sc)


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = Linear(4 * dim, 2 * dim, bias=False)
        self.reduction.weight._clv_want_t = ops.wants_transposed(2 * dim, 4 * dim)
        self.norm = norm_layer(4 * dim)

    @staticmethod
    def merge_gather(x):
        """[B,D,H,W,C] -> [B,D,H/2,W/2,4C] in the reference's concat order (:531-539)."""
        B, D, H, W, C = x.shape
        if (H % 2 == 1) or (W % 2 == 1):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        B, D, H, W, C = x.shape
        # concat order [x(h even,w even), x(h odd,w even), x(h even,w odd), x(h odd,w odd)] = channel block
        # 2 * (w parity) + (h parity): ONE strided copy forward and ONE backward instead of 4 slices + cat
        v = x.view(B, D, H // 2, 2, W // 2, 2, C).permute(0, 1, 2, 4, 5, 3, 6)
        return v.reshape(B, D, H // 2, W // 2, 4 * C)

    def forward(self, x):
        return self.reduction(self.norm(self.merge_gather(x)))

    def forward_pending(self, s, m, sc):
        """Downsample the stream s + sc * m (the last block's pending residual) without materialising it: gather,
        residual add and DropPath factor all ride in the LayerNorm kernel (ops.merge_layer_norm)."""
        B, D, H, W, C = s.shape
        if ops.parity.enabled():
            return self.forward(ops.parity.rnd('stream', s + DropPath.apply_scale(m, sc)))
        if not s.is_cuda or H % 2 or W % 2 or C % 8 or not isinstance(self.norm, LayerNorm):
            return self.forward(s + DropPath.apply_scale(m, sc))
        y = ops.merge_layer_norm(m, self.norm.weight, self.norm.bias, self.norm.eps, residual=s, x_scale=sc)
        return self.reduction(y)


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=(1, 7, 7), mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., norm_layer=LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        self.window_size = window_size
        self.shift_size = tuple(i // 2 for i in window_size)
        self.depth = depth
        self.use_checkpoint = use_checkpoint
        self.blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim=dim, num_heads=num_heads, window_size=window_size,
                                   shift_size=(0, 0, 0) if (i % 2 == 0) else self.shift_size, mlp_ratio=mlp_ratio,
                                   qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                   drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                                   norm_layer=norm_layer, use_checkpoint=use_checkpoint)
            for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        """x bf16 channels-last [B,D,H,W,C] -> [B,D,H',W',C'] (the reference takes/returns B C D H W)."""
        s, m, sc = self.forward_pending(x)
        if self.downsample is not None and hasattr(self.downsample, 'forward_pending'):
            return self.downsample.forward_pending(s, m, sc)
        x = s + DropPath.apply_scale(m, sc)
        if self.downsample is not None:
            x = self.downsample(x)
        return x

    def forward_pending(self, x, branch=None, bscale=None):
        for blk in self.blocks:
            if self.use_checkpoint and torch.is_grad_enabled():
                # the reference wraps both halves of a block in checkpoint.checkpoint (:494-503): keep the block's
                # input, drop its activations, recompute them in the backward.  The DropPath factors are the step's
                # presets (_draw_drop_paths) and the Swin blocks have no dropout, so the recompute is exact; the
                # kernel-side gradient sinks see ONE backward per layer as without it.  preserve_rng_state=False when the
                # block has no dropout (every Clover config: drop_rate 0): the recompute then draws nothing, and saving /
                # restoring the generator state is illegal inside the engine's hipGraph capture.  With
                # drop_rate > 0 the block's nn.Dropout layers DO draw: the recompute must see the forward's generator
                # state or the backward differentiates other masks than the forward applied.
                draws = blk.mlp.drop.p > 0 or blk.attn.proj_drop.p > 0
                x, branch, bscale = _checkpoint.checkpoint(blk.forward_pending, x, branch, bscale, use_reentrant=False,
                                                           preserve_rng_state=draws)
            else:
                x, branch, bscale = blk.forward_pending(x, branch, bscale)
        return x, branch, bscale


class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size=(2, 4, 4), in_chans=3, embed_dim=96, norm_layer=None, stride=(2, 4, 4)):
        super().__init__()
        if tuple(patch_size) != (2, 4, 4) or tuple(stride) != (2, 4, 4) or in_chans != 3:
            raise NotImplementedError('the HIP patch-embed kernel covers patch=stride=(2,4,4), in_chans=3 '
                                      '(every Clover / VideoSwin config)')
        self.patch_size = tuple(patch_size)
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=patch_size, stride=stride)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def pad(self, x):
        _, _, D, H, W = x.size()
        if W % self.patch_size[2] != 0:
            x = F.pad(x, (0, self.patch_size[2] - W % self.patch_size[2]))
        if H % self.patch_size[1] != 0:
            x = F.pad(x, (0, 0, 0, self.patch_size[1] - H % self.patch_size[1]))
        if D % self.patch_size[0] != 0:
            x = F.pad(x, (0, 0, 0, 0, 0, self.patch_size[0] - D % self.patch_size[0]))
        return x

    def tokens(self, x, mask_token=None, vmask=None, want_clean=True):
        """fp32 clip [B,3,T,H,W] -> (clean, masked) bf16 channels-last tokens."""
        x = self.pad(x)
        g, b = (self.norm.weight, self.norm.bias) if self.norm is not None else (None, None)
        eps = self.norm.eps if self.norm is not None else 1e-5
        return ops.patch_embed(x, self.proj.weight, self.proj.bias, g, b, mask_token, vmask, want_clean, eps)

    def tokens_stacked(self, x, mask_token, vmask):
        """fp32 clip [B,3,T,H,W] -> bf16 [2B,T',H',W',C]: clean tokens, then mask-token-blended tokens."""
        x = self.pad(x)
        g, b = (self.norm.weight, self.norm.bias) if self.norm is not None else (None, None)
        eps = self.norm.eps if self.norm is not None else 1e-5
        return ops.patch_embed_stacked(x, self.proj.weight, self.proj.bias, g, b, mask_token, vmask, eps)

    def forward(self, x):
        """Reference contract: [B,3,T,H,W] -> [B,C,T',H',W']."""
        clean, _ = self.tokens(x)
        return clean.permute(0, 4, 1, 2, 3)


@BACKBONES.register_module()
class SwinTransformer3D(nn.Module):
    def __init__(self, pretrained=None, pretrained2d=True, patch_size=(2, 4, 4), stride=(2, 4, 4), in_chans=3,
                 embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=(8, 7, 7), mlp_ratio=4.,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.1,
                 norm_layer=LayerNorm, patch_norm=True, frozen_stages=-1, use_checkpoint=False, mask_token=False):
        super().__init__()
        if norm_layer is nn.LayerNorm:
            norm_layer = LayerNorm
        self.pretrained = pretrained
        self.pretrained2d = pretrained2d
        self.num_layers = len(depths)
        self.embed_dim = embed_dim
        self.patch_norm = patch_norm
        self.frozen_stages = frozen_stages
        self.window_size = tuple(window_size)
        self.patch_size = tuple(patch_size)
        self.fp16_enabled = False
        self.patch_embed = PatchEmbed3D(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, stride=stride,
                                        norm_layer=norm_layer if self.patch_norm else None)
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i_layer in range(self.num_layers):
            self.layers.append(BasicLayer(
                dim=int(embed_dim * 2 ** i_layer), depth=depths[i_layer], num_heads=num_heads[i_layer],
                window_size=self.window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                drop=drop_rate, attn_drop=attn_drop_rate,
                drop_path=dpr[sum(depths[:i_layer]):sum(depths[:i_layer + 1])], norm_layer=norm_layer,
                downsample=PatchMerging if i_layer < self.num_layers - 1 else None, use_checkpoint=use_checkpoint))
        self.num_features = int(embed_dim * 2 ** (self.num_layers - 1))
        self.norm = norm_layer(self.num_features)
        if mask_token:
            self.mask_token = nn.Parameter(torch.zeros(1, self.embed_dim, 1, 1, 1))
            trunc_normal_(self.mask_token, mean=0., std=.02)
        self._freeze_stages()

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.patch_embed.eval()
            for p in self.patch_embed.parameters():
                p.requires_grad = False
        if self.frozen_stages >= 1:
            self.pos_drop.eval()
            for i in range(0, self.frozen_stages):
                m = self.layers[i]
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False

    def inflate_state_dict(self, state_dict):
        """2D Swin -> 3D inflation of a checkpoint state_dict, exactly the reference's arithmetic (:130-181):
        index / mask buffers dropped; patch_embed.proj.weight repeated along time and divided by patch_t; every
        relative_position_bias_table bicubically resized to (2wh-1)(2ww-1) rows if needed and tiled (2wd-1) times
        (a head-count mismatch is tiled unresized, as the reference does, and then fails the shape check of the load)."""
        sd = {k: v for k, v in state_dict.items() if 'relative_position_index' not in k and 'attn_mask' not in k}
        w = sd['patch_embed.proj.weight']
        sd['patch_embed.proj.weight'] = w.unsqueeze(2).repeat(1, 1, self.patch_size[0], 1, 1) / self.patch_size[0]
        own = self.state_dict()
        for k in [k for k in sd if 'relative_position_bias_table' in k]:
            pre = sd[k]
            L1, nH1 = pre.size()
            nH2 = own[k].size(1)
            L2 = (2 * self.window_size[1] - 1) * (2 * self.window_size[2] - 1)
            wd = self.window_size[0]
            if nH1 == nH2 and L1 != L2:
                S1 = int(L1 ** 0.5)
                pre = F.interpolate(pre.permute(1, 0).view(1, nH1, S1, S1),
                                    size=(2 * self.window_size[1] - 1, 2 * self.window_size[2] - 1), mode='bicubic')
                pre = pre.view(nH2, L2).permute(1, 0)
            sd[k] = pre.repeat(2 * wd - 1, 1)
        return sd

    def inflate_weights(self, state_dict):
        """Inflate and load (non-strict; entries whose shape does not fit are reported and skipped, as mmcv's
        load_state_dict does).  -> (missing keys, unexpected keys, shape-mismatched keys)."""
        sd = self.inflate_state_dict(state_dict)
        own = self.state_dict()
        bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
        res = self.load_state_dict({k: v for k, v in sd.items() if k not in bad}, strict=False)
        return list(res.missing_keys), list(res.unexpected_keys), bad

    def init_weights(self, pretrained=None):
        def _init_weights(m):
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if pretrained:
            self.pretrained = pretrained
        if isinstance(self.pretrained, str):
            self.apply(_init_weights)
            ckpt = torch.load(self.pretrained, map_location='cpu')
            sd = ckpt.get('state_dict', ckpt)
            if self.pretrained2d:
                self.inflate_weights(sd)
            else:
                sd = {(k[len('backbone.'):] if k.startswith('backbone.') else k): v for k, v in sd.items()}
                self.load_state_dict(sd, strict=False)
        elif self.pretrained is None:
            self.apply(_init_weights)
        else:
            raise TypeError('pretrained must be a str or None')

    # ---- channels-last core -------------------------------------------------------------
    def _draw_drop_paths(self, B, device):
        """All DropPath factors of this pass from ONE RNG call: [n, B] = bernoulli(keep_i) / keep_i."""
        if not self.training:
            return
        if getattr(self, '_dp_mods', None) is None:
            self._dp_mods = [m for m in self.modules() if isinstance(m, DropPath) and m.drop_prob > 0]
            self._dp_tables = {}
        if not self._dp_mods:
            return
        # One (keep probabilities [n, B], 1 / keep [n, 1]) pair per (device, B), NEVER evicted: a captured hipGraph bakes the
        # address of the table its bernoulli launch reads, and the engine keeps one set of graphs per batch geometry
        # (engine.step -> capture / _activate) — a single slot re-keyed on B freed the table the first geometry's graph
        # still read when a second per-rank B came along (tests/test_engine_gpu.py::test_graphs_per_batch_geometry
        # replays the first geometry after capturing a second with another B).
        key = (str(device), int(B))
        tab = self._dp_tables.get(key)
        if tab is None:
            keep = torch.tensor([1.0 - m.drop_prob for m in self._dp_mods], device=device)[:, None]
            tab = self._dp_tables[key] = (keep.expand(len(self._dp_mods), B).contiguous(), 1.0 / keep)
        scales = torch.bernoulli(tab[0]) * tab[1]                                 # two kernels (was rand, <, cast, /)
        for i, m in enumerate(self._dp_mods):
            m.preset(scales[i])

    # Data-parallel graph mode cuts the backward inside the encoder as well: the stages from CUT_STAGE on hold
    # ~90 % of the encoder's parameters (Swin-T: 25 M of 28 M) but run on 1/16 of the tokens, so their backward is
    # through early and their gradient buckets can travel under the token-heavy backward of stages 0-1.
    CUT_STAGE = 2

    def late_parameters(self):
        """Parameters whose gradients are complete once the backward has come down to the CUT_STAGE input."""
        if self.CUT_STAGE >= self.num_layers - 1:      # no such cut point in _stages (the last stage is not cut)
            return []
        mods = list(self.layers[self.CUT_STAGE:]) + [self.norm]
        return [q for m in mods for q in m.parameters()]

    def _stages(self, x, mid_cut=None):
        self._draw_drop_paths(x.shape[0], x.device)
        x = self.pos_drop(x)
        for i, layer in enumerate(self.layers[:-1]):
            if mid_cut is not None and i == self.CUT_STAGE:
                leaf = x.detach().requires_grad_()
                mid_cut.append((x, leaf))
                x = leaf
            x = layer(x)
        s, m, sc = self.layers[-1].forward_pending(x)    # last stage has no downsample:
        return self.norm(m, residual=s, x_scale=sc)      # its final residual add rides in the norm

    def forward_tokens(self, x, mask=None, mid_cut=None):
        """[B,3,T,H,W] -> channels-last features [B,T',h,w,Cf] (masked pass if `mask` given)."""
        if mask is None:
            t, _ = self.patch_embed.tokens(x)
        else:
            _, t = self.patch_embed.tokens(x, self.mask_token, mask, want_clean=False)
        return self._stages(t, mid_cut)

    def forward_pair(self, x, mask):
        """Clean + masked pass of the pre-training step as ONE 2B-clip pass.
        Returns (clean [B,T',h,w,Cf], masked [B,T',h,w,Cf]) channels-last bf16."""
        B = x.shape[0]
        clean, masked = self.patch_embed.tokens(x, self.mask_token, mask)
        y = self._stages(torch.cat([clean, masked], dim=0))
        return y[:B], y[B:]

    def forward_both(self, x, mask, mid_cut=None):
        """As ``forward_pair`` but returns the one [2B,T',h,w,Cf] tensor (clean clips first, masked clips second).
        mid_cut: optional list that receives the (tensor, detached leaf) pair of the in-encoder backward cut."""
        return self._stages(self.patch_embed.tokens_stacked(x, self.mask_token, mask), mid_cut)

    def forward(self, x, mask=None):
        """Reference contract: [B,3,T,H,W] -> [B,Cf,T',h,w]; with ``mask`` -> (x, w)."""
        y = self.forward_tokens(x, mask).permute(0, 4, 1, 2, 3)
        if mask is not None:
            _, _, T, H, W = (x.shape[0], 0, (x.shape[2] + 1) // 2, (x.shape[3] + 3) // 4, (x.shape[4] + 3) // 4)
            w = mask_blend_weight(mask, T, H, W).type_as(y)
            return y, w
        return y

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
