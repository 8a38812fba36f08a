"""Closed-form (RNG-free) weights and inputs shared by the golden generator and the
parity tests, so fixtures only need to hold OUTPUTS (SURVEY.md §8c G-swin note).

Every tensor is a deterministic function of (name, shape): a splitmix64 hash of
(crc32(name), flat index) mapped to uniform [-1, 1) (floats, full-rank, RNG-free) or to an
integer range.  Magnitudes follow real initialisations (weights ~ 1/sqrt(fan_in), embeddings
0.1, LayerNorm gains 1 +- 0.1) so that attention logits stay O(1-10) as in a trained model.
Nothing here depends on torch's RNG or on the reference.
"""
import zlib

import numpy as np
import torch

TINY_BERT = dict(vocab_size=1024, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                 intermediate_size=512, max_position_embeddings=64, type_vocab_size=2,
                 layer_norm_eps=1e-12)

AUX = ['token_ids', 'segment_ids', 'input_mask', 'mlm_label', 'v_token_mask']


def tiny_model_cfg(drop=0.0):
    """BASELINE config 1 in the reference's own config format
    (configs/exp_local/pretrain_webvid_cc3m.py:22-112, shrunk): 2-stage tiny
    SwinTransformer3D + 2-layer BERT-tiny + 2-layer fusion, all five losses on.
    ``bert_config`` is an extra kwarg the reference classes swallow via **kwargs."""
    return dict(
        type='CloverPretrain', freeze_stage=None, separate_test=True, use_Cmask=True,
        backbone=dict(type='SwinTransformer3D', patch_size=(2, 4, 4), stride=(2, 4, 4), embed_dim=48,
                      depths=[2, 2], num_heads=[3, 6], window_size=(8, 7, 7), drop_path_rate=drop,
                      mask_token=True, pretrained2d=False, pretrained=None),
        freeze_text_backbone=None, text_vocab_size=1024,
        mm_backbone=dict(type='CrossModalTransformerFromPretrained', use_text_cls=True, use_prompt=False,
                         pretrained_model='bert-base-uncased', num_hidden_layers=2, img_in_size=96,
                         hidden_size=128, num_frames=2, spacial_tokens=14 * 14, token_types=2,
                         layer_norm_eps=1e-12, word_pos_start=False, bert_config=dict(TINY_BERT)),
        text_backbone=dict(type='BertFromPretrained', num_hidden_layers=2, bert_config=dict(TINY_BERT)),
        cls_head=None,
        ssl_head=dict(type='NCEHeadForMM', visual_in_channels=96, text_in_channels=128, img_hidden_dim=256,
                      vts_embed_dim=128, ln=True, spatial_type='avg', text_agg_type='cls', dropout_ratio=0),
        mlm_head=dict(type='MLMHead', hidden_size=128, vocab_size=1024),
        mlm_ssl_head=dict(
            V=dict(type='NCEHeadForVision', visual_in_channels=128, cross_in_channels=128, hidden_dim=128,
                   ln=True, vts_embed_dim=128, dropout_ratio=0),
            T=dict(type='NCEHeadForText', cross_in_channels=128, vts_embed_dim=128, text_bn=False,
                   dropout_ratio=0.1)),
        mlm_loss=dict(type='SoftmaxFocalLossMultiClass', gamma=2.0),
        loss_type=dict(type='CrossEntropyLoss'),
        ssl_loss=dict(type='ExclusiveNCEwithRankingLoss', temperature=0.05, use_rank=True, use_rank_ttm=True,
                      use_rank_trtm=False, margin_ttm=5., margin_trtm=10.),
        symmetry_rank=True,
        train_cfg=dict(aux_info=list(AUX)))


def tiny_finetune_cfg(cos_sim=True, temperature=0.05):
    """The retrieval fine-tuning model (configs/exp_local/finetune_msrvtt_retrieval.py:23-74) at the tiny widths
    of ``tiny_model_cfg``: same encoders and ssl_head (so the same closed-form weights by name), NormSoftmaxLoss."""
    base = tiny_model_cfg()
    return dict(
        type='CloverFinetune', freeze_stage=None, separate_test=True, backbone=base['backbone'],
        freeze_text_backbone=None, text_vocab_size=1024, mm_backbone=base['mm_backbone'],
        text_backbone=base['text_backbone'], cls_head=None, task='retrieval', ssl_head=base['ssl_head'],
        itm_head=None, loss_type=dict(type='NormSoftmaxLoss', cos_sim=cos_sim, temperature=temperature),
        train_cfg=dict(aux_info=['token_ids', 'segment_ids', 'input_mask']),
        test_cfg=dict(feature_extraction=False))


def oracle_cfg_from(model_cfg, bert_cfg=TINY_BERT):
    """The oracle's compact cfg derived from a reference-format model cfg."""
    bb = {k: v for k, v in model_cfg['backbone'].items()
          if k in ('patch_size', 'embed_dim', 'depths', 'num_heads', 'window_size', 'mlp_ratio', 'patch_norm')}
    return dict(
        backbone=bb,
        bert=dict(num_hidden_layers=model_cfg['text_backbone']['num_hidden_layers'],
                  num_attention_heads=bert_cfg['num_attention_heads'],
                  layer_norm_eps=model_cfg['text_backbone'].get('layer_norm_eps', 1e-12)),
        fusion=dict(num_hidden_layers=model_cfg['mm_backbone']['num_hidden_layers'],
                    num_attention_heads=bert_cfg['num_attention_heads'],
                    layer_norm_eps=model_cfg['mm_backbone'].get('layer_norm_eps', 1e-12)),
        temperature=model_cfg['ssl_loss']['temperature'], margin=model_cfg['ssl_loss']['margin_ttm'],
        gamma=model_cfg['mlm_loss']['gamma'], vocab=model_cfg['text_vocab_size'])


def _splitmix(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _hash_u64(name, n):
    with np.errstate(over='ignore'):
        return _splitmix(np.arange(n, dtype=np.uint64) + np.uint64(zlib.crc32(name.encode())) * np.uint64(1000003))


def cf_float(name, shape, scale=1.0, offset=0.0):
    """offset + scale * U[-1, 1), value i = hash(name, i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (_hash_u64(name, n) >> np.uint64(11)).astype(np.float64) / float(1 << 53)      # [0, 1)
    v = offset + scale * (2.0 * u - 1.0)
    return torch.from_numpy(v.astype(np.float32).reshape(shape))


def cf_int(name, shape, lo, hi):
    n = int(np.prod(shape))
    h = _hash_u64(name + '#int', n)
    v = lo + (h % np.uint64(hi - lo)).astype(np.int64)
    return torch.from_numpy(v.reshape(shape))


def cf_param(name, shape):
    """Closed-form value for a parameter, by the role its name/shape implies."""
    shape = tuple(shape)
    leaf = name.rsplit('.', 1)[-1]
    if len(shape) == 1 and leaf == 'weight':            # LayerNorm gains
        return cf_float(name, shape, 0.1, 1.0)
    if len(shape) == 1:                                 # biases / LN shifts
        return cf_float(name, shape, 0.05)
    if 'relative_position_bias_table' in name:
        return cf_float(name, shape, 0.5)
    if 'mask_token' in name or 'vis_space_pos' in name or 'vis_tempor_pos' in name:
        return cf_float(name, shape, 0.1)
    if 'embeddings' in name:                            # embedding tables
        return cf_float(name, shape, 0.1)
    fan_in = int(np.prod(shape[1:]))
    return cf_float(name, shape, 1.2 / np.sqrt(fan_in))


def cf_state(manifest):
    """manifest: {name: shape}. Integer buffers (relative_position_index) are skipped."""
    return {k: cf_param(k, s) for k, s in manifest.items() if 'relative_position_index' not in k}


def cf_batch(B, frames=4, size=112, L=16, vocab=1024, tag='b', n_pad=3):
    """Synthetic batch in the reference's data_batch format."""
    imgs = cf_float(f'{tag}.imgs', (B, 1, 3, frames, size, size), 1.7)
    ids = cf_int(f'{tag}.ids', (B, 1, L), 5, vocab)
    ids[:, :, 0] = 101
    input_mask = torch.ones(B, 1, L, dtype=torch.long)
    for b in range(B):
        npad = (b * 2 + n_pad) % (L // 2)
        if npad:
            input_mask[b, :, L - npad:] = 0
            ids[b, :, L - npad:] = 0
        ids[b, :, L - npad - 1] = 102
    mlm_label = torch.full((B, 1, L), -100, dtype=torch.long)
    sel = cf_int(f'{tag}.sel', (B, 1, L), 0, 10) < 3
    sel[:, :, 0] = False
    sel &= (input_mask == 1) & (ids != 102)
    sel[:, :, 2] = True                                   # at least one masked row per sample
    mlm_label[sel] = ids[sel]
    token_ids = ids.clone()
    token_ids[sel] = 103
    vm = torch.zeros(B, 1, 7, 7, dtype=torch.long)
    for b in range(B):
        r0, c0 = (b * 3) % 4, (b * 5) % 3
        vm[b, :, r0:r0 + 3, c0:c0 + 4] = 1
    return dict(imgs=imgs, label=torch.zeros(B, dtype=torch.long), token_ids=token_ids,
                segment_ids=torch.zeros_like(token_ids), input_mask=input_mask,
                mlm_label=mlm_label, v_token_mask=vm)


def inflate_checkpoint_2d():
    """A synthetic 2-D Swin checkpoint (closed-form values) for the tiny backbone: same-size tables (13 x 13) in
    stage 0, a 23 x 23 table (window 12) in stage 1 block 0 to exercise the bicubic resize, plus the keys the
    inflation must drop.  Shared by make_goldens.py (fed to the reference) and tests/test_host_cpu.py."""
    C = 48
    sd = {
        'patch_embed.proj.weight': cf_float('inflate.proj', (C, 3, 4, 4), 0.1),
        'patch_embed.proj.bias': cf_float('inflate.projb', (C,), 0.1),
        'layers.0.blocks.0.attn.relative_position_bias_table': cf_float('inflate.t00', (13 * 13, 3), 0.5),
        'layers.0.blocks.1.attn.relative_position_bias_table': cf_float('inflate.t01', (13 * 13, 3), 0.5),
        'layers.1.blocks.0.attn.relative_position_bias_table': cf_float('inflate.t10', (23 * 23, 6), 0.5),
        'layers.1.blocks.1.attn.relative_position_bias_table': cf_float('inflate.t11', (13 * 13, 5), 0.5),   # nH mismatch: skipped
        'layers.0.blocks.0.attn.relative_position_index': torch.zeros(49, 49, dtype=torch.long),
        'layers.0.blocks.1.attn_mask': torch.zeros(4, 49, 49),
        'layers.0.blocks.0.attn.qkv.weight': cf_float('inflate.qkv', (3 * C, C), 0.05),
        'norm.weight': cf_float('inflate.norm', (2 * C,), 0.2) + 1.0,
    }
    return sd
