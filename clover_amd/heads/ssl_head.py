"""Contrastive projection heads, registered under the reference's names/kwargs
(mmaction/models/heads/ssl_head.py:9-297).  [B, D]-sized GEMMs kept in fp32 (see
nn.LinearFP32) + the HIP LayerNorm/GELU kernels in their fp32-storage form; outputs fp32
(the loss is @force_fp32)."""
import torch
import torch.nn as nn

from .. import ops
from ..builder import HEADS
from ..nn import GELU, BatchNorm1d, LayerNorm
from ..nn import LinearFP32 as Linear


def _init_head(module):
    for _, m in module.named_modules():
        if isinstance(m, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
        elif isinstance(m, (nn.BatchNorm1d, nn.LayerNorm)):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)
        if isinstance(m, nn.Linear) and m.bias is not None:
            m.bias.data.zero_()


def _norm(dim, ln):
    """nn.LayerNorm (ln=True, every published config) or nn.BatchNorm1d (ssl_head.py:52,56,175-186) of a head layer."""
    return LayerNorm(dim) if ln else BatchNorm1d(dim)


def _per_pass(fn, x, passes, order, live=None):
    """Apply fn to the row blocks of a batch that stacks `passes` forward passes of the reference, one block at a time in
    the reference's call order: BatchNorm statistics (and their running averages) are per CALL there — the clean and the
    masked clips, the un-masked and the masked captions each go through the head on their own (:102, :150, :159).
    live: the blocks the reference really runs under the current ablation switches (None = all).  The others are still
    computed (their slot of the packed embeddings exists), but with the BatchNorm layers' running statistics frozen: the
    reference never shows those inputs to the layer."""
    from ..nn import BatchNorm1d as _BN
    blocks = list(x.chunk(passes, dim=0))
    outs = [None] * passes
    for i in (order if order is not None else range(passes)):
        if live is not None and i not in live:
            with _BN.frozen_stats():
                outs[i] = fn(blocks[i])
        else:
            outs[i] = fn(blocks[i])
    return torch.cat(outs, dim=0)


@HEADS.register_module()
class NCEHeadForMM(nn.Module):
    def __init__(self, visual_in_channels, text_in_channels, img_hidden_dim, vts_embed_dim, spatial_type='avg',
                 text_agg_type='avg', ln=False, text_bn=False, dropout_ratio=0.1, init_std=0.01, **kwargs):
        super().__init__()
        self.vis_in_channels = visual_in_channels
        self.text_in_channels = text_in_channels
        self.spatial_type = spatial_type
        self.dropout_ratio = dropout_ratio
        self.init_std = init_std
        self.img_hidden_dim = img_hidden_dim
        self.vts_embed_dim = vts_embed_dim
        self.fp16_enabled = False
        self.ln = ln
        self.dropout = nn.Dropout(p=dropout_ratio) if dropout_ratio != 0 else None
        self.img_projector = nn.Sequential(
            Linear(visual_in_channels, img_hidden_dim), _norm(img_hidden_dim, ln), GELU(),
            Linear(img_hidden_dim, vts_embed_dim), _norm(vts_embed_dim, ln))
        if text_bn:                                          # :58-64 (the indices of the Sequential are state_dict keys)
            self.text_projector = nn.Sequential(
                Linear(text_in_channels, text_in_channels), BatchNorm1d(text_in_channels), GELU(),
                Linear(text_in_channels, vts_embed_dim))
        else:
            self.text_projector = nn.Sequential(
                Linear(text_in_channels, text_in_channels), GELU(), Linear(text_in_channels, vts_embed_dim))
        self.text_bn = text_bn
        self.init_weights()
        self.text_agg_type = text_agg_type

    def init_weights(self):
        _init_head(self)

    def forward(self, img, text, text_mask=None, token_ids=None):
        return self.forward_vision(img), self.forward_text(text, text_mask, token_ids)

    def forward_vision(self, img, channels_last=False, passes=1, order=None, live=None):
        """img [N,C,T,h,w] (reference layout) or [N,T,h,w,C] with channels_last=True -> [N,vts] fp32.
        passes > 1: the batch stacks that many forward passes of the reference (the recognizer's doubled clean + masked
        pass); a BatchNorm projector then runs per pass (see _per_pass), a LayerNorm one does not care."""
        if self.spatial_type == 'avg':
            img = ops.token_mean(img) if channels_last else img.float().mean(dim=(2, 3, 4))
        else:
            raise NotImplementedError('spatial_type other than avg')
        if self.dropout is not None:
            img = self.dropout(img)
        if passes > 1 and not self.ln:
            return _per_pass(self.img_projector, img, passes, order, live).float()
        return self.img_projector(img).float()

    def forward_text(self, text, text_mask=None, token_ids=None, passes=1, order=None, live=None):
        if self.text_agg_type == 'avg':
            text_mask = torch.where(token_ids != 102, text_mask, torch.zeros_like(text_mask))
            text = text[:, 1:].float()
            text_mask = text_mask[:, 1:]
            text = (text * text_mask.unsqueeze(-1)).sum(1) / text_mask.sum(1, keepdim=True)
        elif self.text_agg_type == 'cls':
            text = text[:, 0]
        elif self.text_agg_type == 'max':
            text_mask = torch.where(token_ids != 102, text_mask, torch.zeros_like(text_mask))
            text = (text[:, 1:].float() * text_mask[:, 1:].unsqueeze(-1)).max(dim=1)[0]
        if passes > 1 and self.text_bn:
            return _per_pass(self.text_projector, text, passes, order, live).float()
        return self.text_projector(text).float()


@HEADS.register_module()
class NCEHeadForVision(nn.Module):
    def __init__(self, cross_in_channels=768, visual_in_channels=1024, hidden_dim=768, vts_embed_dim=768,
                 dropout_ratio=0.1, ln=False, init_std=0.01, **kwargs):
        super().__init__()
        self.cross_in_channels = cross_in_channels
        self.visual_in_channels = visual_in_channels
        self.vts_embed_dim = vts_embed_dim
        self.hidden_dim = hidden_dim
        self.dropout_ratio = dropout_ratio
        self.ln = ln
        self.dropout = nn.Dropout(p=dropout_ratio) if dropout_ratio != 0 else None
        self.img_fc1 = Linear(visual_in_channels, hidden_dim * 2)
        self.img_bn1 = _norm(hidden_dim * 2, ln)
        self.img_act = GELU()
        self.img_fc2 = Linear(hidden_dim * 2, vts_embed_dim)
        self.img_bn2 = _norm(vts_embed_dim, ln)
        self.init_weights()

    def init_weights(self):
        _init_head(self)

    def forward(self, img):
        """img [b, tokens, C] -> mean over tokens; a 2-D CLS row [b, C] is taken as-is.

        Deviation D1: the reference calls ``img.mean(dim=1)`` on the 2-D row
        ``t_last_hidden_state[:, 0]`` (multimodal_transformer_pretrain.py:148-149), which
        collapses it to [b] and crashes in ``img_fc1``.  The only shape-consistent reading —
        also what the symmetric NCEHeadForText does — is to treat the row as [b,1,C]."""
        if img.dim() == 3:
            img = img.float().mean(dim=1)
        if self.dropout is not None:
            img = self.dropout(img)
        return self.img_bn2(self.img_fc2(self.img_act(self.img_bn1(self.img_fc1(img))))).float()


@HEADS.register_module()
class NCEHeadForText(nn.Module):
    def __init__(self, cross_in_channels=768, vts_embed_dim=768, dropout_ratio=0.1, text_bn=False, **kwargs):
        super().__init__()
        self.cross_in_channels = cross_in_channels
        self.vts_embed_dim = vts_embed_dim
        self.dropout_ratio = dropout_ratio
        self.fp16_enabled = False
        self.dropout = nn.Dropout(p=dropout_ratio) if dropout_ratio != 0 else None
        self.text_bn = text_bn
        self.fc1 = Linear(cross_in_channels, cross_in_channels)
        self.bn = BatchNorm1d(cross_in_channels) if text_bn else None
        self.act = GELU()
        self.fc2 = Linear(cross_in_channels, vts_embed_dim)
        self.init_weights()

    def init_weights(self):
        _init_head(self)

    def forward(self, mask_word_feat):
        x = self.fc1(mask_word_feat)
        if self.bn is not None:
            x = self.bn(x)
        x = self.act(x)
        if self.dropout is not None:
            x = self.dropout(x)
        return self.fc2(x).float()
