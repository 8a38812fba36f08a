"""CPU-only checks of the host side: registry / plugin API, state_dict compatibility, the
index/geometry helpers against the reference goldens (bit-exact), the C ABI surface, and the
no-CPU-fallback rule."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest
import torch

import closed_form as cf
import gutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_builds_reference_config_and_state_dict_matches_reference():
    import clover_amd
    from clover_amd.builder import MODELS
    for name in ['SwinTransformer3D', 'BertFromPretrained', 'CrossModalTransformerFromPretrained', 'NCEHeadForMM',
                 'NCEHeadForVision', 'NCEHeadForText', 'MLMHead', 'ExclusiveNCEwithRankingLoss',
                 'SoftmaxFocalLossMultiClass', 'CrossEntropyLoss', 'CloverPretrain']:
        assert name in MODELS, name
    m = clover_amd.build_model(cf.tiny_model_cfg())
    sd = m.state_dict()
    man = gutil.manifest()                     # names/shapes of the REFERENCE model's state_dict
    assert set(sd) == set(man)
    for k, shp in man.items():
        assert list(sd[k].shape) == shp, k
    assert m.aux_info == cf.AUX and m.fp16_enabled is False
    with pytest.raises(KeyError):
        clover_amd.build_backbone(dict(type='NoSuchBackbone'))
    with pytest.raises(ValueError):
        m(torch.zeros(1), None, return_loss=True)          # 'Label should not be None.'


FS = [(4, 56, 56), (8, 56, 56), (16, 56, 56), (2, 28, 28), (2, 14, 14), (4, 7, 7), (16, 7, 7), (16, 14, 14), (3, 10, 12)]


@pytest.mark.parametrize('fs', FS)
def test_window_geometry_bit_exact(fs):
    from clover_amd.backbones import swin_transformer_3d as S
    g = gutil.load('g_idx.npz')
    tag = '%d_%d_%d' % fs
    ws, ss = S.get_window_size(fs, (8, 7, 7), (4, 3, 3))
    assert list(ws) + list(ss) == g['gws_' + tag].tolist()
    ws2, ss2, rid = S.window_geometry(fs, (8, 7, 7), (4, 3, 3), 'cpu')
    assert (ws2, ss2) == (ws, ss)
    mask = g['mask_' + tag]
    if any(ss):
        r = rid.numpy()
        assert r.dtype == np.int32
        assert np.array_equal((r[:, :, None] != r[:, None, :]).astype(np.int8), mask)
    else:
        assert rid is None and not mask.any()


def _tok_row(D, H, W, ws, ss, wi, n):
    """Python transcription of the kernel's token addressing (attention.hip tok_row)."""
    nWh, nWw = H // ws[1], W // ws[2]
    wz, wr = divmod(wi, nWh * nWw)
    wy, wx = divmod(wr, nWw)
    tz, tr = divmod(n, ws[1] * ws[2])
    ty, tx = divmod(tr, ws[2])
    d = (wz * ws[0] + tz + ss[0]) % D
    h = (wy * ws[1] + ty + ss[1]) % H
    w = (wx * ws[2] + tx + ss[2]) % W
    return (d * H + h) * W + w


@pytest.mark.parametrize('fs', [(4, 56, 56), (16, 14, 14), (2, 28, 28), (4, 7, 7)])
def test_kernel_token_addressing_matches_roll_partition(fs):
    from clover_amd.backbones import swin_transformer_3d as S
    g = gutil.load('g_idx.npz')
    ws, ss = S.get_window_size(fs, (8, 7, 7), (4, 3, 3))
    part = g['part_%d_%d_%d' % fs]                      # reference roll(-shift) + window_partition of arange
    nW, N = part.shape
    for wi in list(range(0, nW, max(1, nW // 7))) + [nW - 1]:
        rows = [_tok_row(fs[0], fs[1], fs[2], ws, ss, wi, n) for n in range(N)]
        assert rows == part[wi].tolist()


def test_relative_position_index_and_blend_and_merge_order():
    from clover_amd.backbones import swin_transformer_3d as S
    g = gutil.load('g_idx.npz')
    for ws in [(8, 7, 7), (4, 7, 7), (2, 7, 7)]:
        assert np.array_equal(S.build_relative_position_index(ws).numpy(), g['rpi_%d_%d_%d' % ws])
    w = S.mask_blend_weight(torch.from_numpy(g['blend_mask']), 2, 28, 28)
    assert np.array_equal(w.numpy().astype(np.int8), g['blend_w'])
    cat = S.PatchMerging.merge_gather(torch.from_numpy(g['merge_in']))
    assert np.array_equal(cat.numpy(), g['merge_cat'])
    tid = torch.from_numpy(g['ssl_token_ids'])[:, 0]
    lab = torch.from_numpy(g['ssl_mlm_label'])[:, 0]
    assert np.array_equal(torch.where(lab == -100, tid, lab).numpy(), g['ssl_ids'])


def test_c_abi_exports_every_declared_symbol():
    from clover_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'clover_hip.h')).read()
    declared = set(re.findall(r'\b(clv_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # both builds of the one ABI (fp16 element type = the default, bf16): every declared symbol, the same version, and each
    # says which element type it computes in (no compute call: there is no GPU here)
    pkg = os.path.dirname(_lib.LIB_PATH)
    for fname, half in (('libclover_hip_f16.so', 1), ('libclover_hip.so', 0)):
        so = ctypes.CDLL(os.path.join(pkg, fname))
        for name in declared:
            assert hasattr(so, name), (fname, name)
        assert so.clv_abi_version() == _lib.ABI_VERSION, fname
        assert so.clv_half_type() == half, fname
    assert os.path.basename(_lib.LIB_PATH) == ('libclover_hip_f16.so' if _lib.HALF_F16 else 'libclover_hip.so')
    assert _lib.lib().clv_abi_version() == _lib.ABI_VERSION
    from clover_amd import ops
    assert ops.OPTIM_STATE_BYTES == int(re.search(r'#define CLV_OPTIM_STATE_BYTES (\d+)', hdr).group(1))
    assert ctypes.sizeof(_lib.ClvAttnGeom) == 23 * 4 + 4 + 8 + 8  # 21 int32 + scale + dropout_p, padding, dbias_index + work pointers
    assert ctypes.sizeof(_lib.ClvWgradEntry) == 88                # ... + overwrite / pad (first-touch gradient sinks)
    assert ctypes.sizeof(_lib.ClvLnExtra) == 80                   # ... + q8 / qscale pointers (round 3)


def test_no_cpu_fallback_and_no_oracle_in_product():
    from clover_amd import ops
    x = torch.randn(4, 96)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.layer_norm(x, torch.ones(96), torch.zeros(96))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.gelu(x)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.linear(x, torch.randn(8, 96), None)
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'clover_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_optimizer_param_groups_and_lr_schedule():
    import clover_amd
    from clover_amd.engine import cosine_lr, paramwise_weight_decay
    m = clover_amd.build_model(cf.tiny_model_cfg())
    wd = paramwise_weight_decay(m, 0.005, 0.0, 0.0, {'absolute_pos_embed': dict(decay_mult=0.),
                                                    'relative_position_bias_table': dict(decay_mult=0.)})
    assert wd['backbone.layers.0.blocks.0.attn.relative_position_bias_table'] == 0.0      # custom key
    assert wd['backbone.layers.0.blocks.0.norm1.weight'] == 0.0                            # norm layer
    assert wd['backbone.layers.0.blocks.0.attn.qkv.bias'] == 0.0                           # bias
    assert wd['backbone.layers.0.blocks.0.attn.qkv.weight'] == 0.005
    assert wd['text_backbone.bert.embeddings.word_embeddings.weight'] == 0.005
    assert wd['multimodal_backbone.vis_space_pos'] == 0.005
    base = 1e-3
    assert abs(cosine_lr(base, 0, 100) - base) < 1e-12
    assert abs(cosine_lr(base, 100, 100) - base * 1e-3) < 1e-12
    assert abs(cosine_lr(base, 50, 100) - (base * 1e-3 + 0.5 * (base - base * 1e-3))) < 1e-12
    lr0 = cosine_lr(base, 0, 1000, warmup_iters=10, warmup_ratio=0.001)
    assert abs(lr0 - cosine_lr(base, 0, 1000) * 0.001) < 1e-12
    assert cosine_lr(base, 10, 1000, warmup_iters=10) == cosine_lr(base, 10, 1000)


def test_paramwise_lr_mult_and_fractional_decay():
    """mmcv DefaultOptimizerConstructor: custom_keys carry lr_mult AND decay_mult (any value), bias_lr_mult applies to
    non-norm biases; a custom-key hit switches the norm / bias rules off."""
    import clover_amd
    from clover_amd.engine import paramwise_options
    m = clover_amd.build_model(cf.tiny_model_cfg())
    o = paramwise_options(m, 0.01, dict(norm_decay_mult=0.0, bias_decay_mult=0.0, bias_lr_mult=2.0, custom_keys={
        'backbone.layers': dict(lr_mult=0.1, decay_mult=0.5), 'relative_position_bias_table': dict(decay_mult=0.)}))
    assert o['backbone.layers.0.blocks.0.attn.relative_position_bias_table'] == (0.0, 1.0)   # longest key wins
    assert o['backbone.layers.0.blocks.0.attn.qkv.weight'] == (0.005, 0.1)
    assert o['backbone.layers.0.blocks.0.attn.qkv.bias'] == (0.005, 0.1)          # custom key: bias rule is off
    assert o['backbone.layers.0.blocks.0.norm1.weight'] == (0.005, 0.1)           # ... and the norm rule
    assert o['text_backbone.bert.encoder.layer.0.output.dense.bias'] == (0.0, 2.0)
    assert o['text_backbone.bert.encoder.layer.0.output.LayerNorm.bias'] == (0.0, 1.0)
    assert o['text_backbone.bert.encoder.layer.0.output.dense.weight'] == (0.01, 1.0)


def test_lr_follows_runner_iter_in_two_loader_mode():
    """the reference's LR hook reads runner.iter, which the two-loader runner advances once per batch
    INDEX (clover_runner.py:76-91), so both loaders' steps of an index share one LR, the first step uses iter 0 and
    warm-up lasts warmup_epochs x len(longest loader) indices.  Checked against mmcv's formula written out."""
    import math
    from clover_amd.runner import CloverRunner, LrUpdaterHook

    class Stepper(_FakeStepper):
        def __init__(self):
            super().__init__()
            self.lrs = []

        def set_lr(self, lr):
            self._lr = lr

        def train_step(self, batch, opt):
            self.lrs.append(self._lr)
            return super().train_step(batch, opt)
    A, B = [f'a{i}' for i in range(6)], [f'b{i}' for i in range(4)]
    base, epochs, warm_epochs = 1e-3, 3, 1
    st = Stepper()
    r = CloverRunner(st, max_epochs=epochs)
    r.register_hook(LrUpdaterHook(base, min_lr_ratio=1e-3, warmup='linear', warmup_iters=warm_epochs,
                                  warmup_ratio=0.001, warmup_by_epoch=True))
    r.run([A, B], [('train', 1)], epochs)
    max_iters, warm = epochs * len(A), warm_epochs * len(A)

    def mmcv_lr(it):
        end = base * 1e-3
        lr = end + 0.5 * (base - end) * (1 + math.cos(math.pi * it / max_iters))          # annealing_cos
        if it < warm:
            k = (1 - it / warm) * (1 - 0.001)                                                # get_warmup_lr 'linear'
            lr = lr * (1 - k)
        return lr
    want = [mmcv_lr(i) for i in range(max_iters) for _ in range(2)]                         # two steps per index
    assert len(st.lrs) == len(want) == 36
    assert max(abs(a - b) for a, b in zip(st.lrs, want)) < 1e-15
    assert st.lrs[0] == mmcv_lr(0) and st.lrs[0] < base * 2e-3                              # iter 0, not 1
    assert st.lrs[2 * warm] == mmcv_lr(warm) and st.lrs[-1] > base * 1e-3                   # never clamped early


def test_bert_loader_maps_legacy_checkpoint_names(tmp_path, monkeypatch):
    """the stock bert-base-uncased file names LayerNorm params gamma / beta and the MLM output bias
    ``cls.predictions.bias``; from_pretrained (bert_from_hugface.py:13-15, mlm_itm_head.py:33-35) renames / ties
    them.  A fake checkpoint in the OLD naming must initialise every parameter; a partial one must raise."""
    from clover_amd.backbones.bert_from_hugface import BertFromPretrained
    from clover_amd.backbones.cross_transformer import CrossModalTransformerFromPretrained
    from clover_amd.heads.mlm_itm_head import MLMHead
    bc = dict(vocab_size=50, hidden_size=16, num_hidden_layers=2, num_attention_heads=2, intermediate_size=32,
              max_position_embeddings=24)
    src = BertFromPretrained(None, bert_config=bc, num_hidden_layers=2)
    g = torch.Generator().manual_seed(3)
    new = {'bert.' + k: torch.randn(v.shape, generator=g) for k, v in src.bert.state_dict().items()}
    new.update({'cls.predictions.transform.dense.weight': torch.randn(16, 16, generator=g),
                'cls.predictions.transform.dense.bias': torch.randn(16, generator=g),
                'cls.predictions.transform.LayerNorm.weight': torch.randn(16, generator=g),
                'cls.predictions.transform.LayerNorm.bias': torch.randn(16, generator=g),
                'cls.predictions.bias': torch.randn(50, generator=g)})                     # no decoder.* keys at all
    old = {k.replace('LayerNorm.weight', 'LayerNorm.gamma').replace('LayerNorm.bias', 'LayerNorm.beta'): v
           for k, v in new.items()}
    assert any('gamma' in k for k in old)
    d = tmp_path / 'bert-base-uncased'
    d.mkdir()
    import json
    (d / 'config.json').write_text(json.dumps(bc))
    torch.save(old, str(d / 'pytorch_model.bin'))
    monkeypatch.chdir(tmp_path)                             # the MLM head's hard-coded 'bert-base-uncased' (:33)
    text = BertFromPretrained('bert-base-uncased', num_hidden_layers=2)
    for k, v in text.bert.state_dict().items():
        assert torch.equal(v, new['bert.' + k]), k
    fuse = CrossModalTransformerFromPretrained('bert-base-uncased', hidden_size=16, img_in_size=16, num_frames=2,
                                               spacial_tokens=4, num_hidden_layers=1, use_text_cls=True)
    assert torch.equal(fuse.bert_encoder.layer[0].output.LayerNorm.weight,
                       new['bert.encoder.layer.0.output.LayerNorm.weight'])
    assert torch.equal(fuse.bert_embedding.LayerNorm.bias, new['bert.embeddings.LayerNorm.bias'])
    head = MLMHead(16, 50)
    assert torch.equal(head.predictions.decoder.bias, new['cls.predictions.bias'])          # tied bias
    assert torch.equal(head.predictions.decoder.weight, new['bert.embeddings.word_embeddings.weight'])   # tied weight
    assert torch.equal(head.predictions.transform.LayerNorm.weight, new['cls.predictions.transform.LayerNorm.weight'])
    # a checkpoint that lacks a parameter must not load silently
    broken = {k: v for k, v in old.items() if 'layer.1.output.LayerNorm.beta' not in k}
    torch.save(broken, str(d / 'pytorch_model.bin'))
    with pytest.raises(RuntimeError, match='not found in the checkpoint'):
        BertFromPretrained('bert-base-uncased', num_hidden_layers=2)


def test_lazy_log_vars_and_parse_losses():
    from clover_amd.recognizers.base import BaseRecognizer, LazyLogVars
    lv = LazyLogVars(['a_loss', 'b'], torch.tensor([1.5, 2.0]))
    assert list(lv.keys()) == ['a_loss', 'b'] and lv['a_loss'] == 1.5 and dict(lv.items())['b'] == 2.0

    class R(BaseRecognizer):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.lazy_log_vars = False

        def forward_train(self, *a, **k):
            pass

        def forward_test(self, *a, **k):
            pass
    loss, log_vars = R()._parse_losses({'mlm_loss': torch.tensor(1.0), 'acc': torch.tensor(0.5),
                                        'x_loss': [torch.tensor([1.0, 3.0])]})
    assert float(loss) == 3.0 and log_vars['loss'] == 3.0 and log_vars['acc'] == 0.5     # 'acc' not summed


def test_swin_2d_to_3d_inflation_matches_reference():
    """SwinTransformer3D.inflate_weights vs the reference's own function run on the same synthetic 2-D
    checkpoint (tests/golden/make_goldens.py::gen_inflate): patch-embed repeat / patch_t, table tiling (2wd-1),
    bicubic 23x23 -> 13x13 resize, dropped buffers, head-count mismatch skipped at load."""
    import clover_amd
    from clover_amd.builder import build_backbone
    g = gutil.load('g_inflate.npz')
    cfg = cf.tiny_model_cfg()['backbone']
    bb = build_backbone(dict(cfg))
    sd = bb.inflate_state_dict(cf.inflate_checkpoint_2d())
    assert sorted(sd) == sorted(str(k) for k in g['__keys__'])
    for k in sd:
        ref = g[k]
        assert tuple(sd[k].shape) == ref.shape, k
        assert np.abs(sd[k].numpy() - ref).max() <= 1e-6 * max(1.0, np.abs(ref).max()), k
    before = bb.layers[1].blocks[1].attn.relative_position_bias_table.detach().clone()
    missing, unexpected, bad = bb.inflate_weights(cf.inflate_checkpoint_2d())
    assert bad == ['layers.1.blocks.1.attn.relative_position_bias_table'] and not unexpected
    assert torch.equal(bb.layers[1].blocks[1].attn.relative_position_bias_table, before)       # left untouched
    assert torch.allclose(bb.layers[1].blocks[0].attn.relative_position_bias_table,
                          torch.from_numpy(g['layers.1.blocks.0.attn.relative_position_bias_table']), atol=1e-6)
    assert torch.allclose(bb.patch_embed.proj.weight, torch.from_numpy(g['patch_embed.proj.weight']), atol=1e-7)


# ----------------------------------------------------------------------------- runner / config
def test_config_base_merge_and_cfg_options(tmp_path):
    from clover_amd.runner import Config, parse_cfg_options, scaled_lr
    (tmp_path / 'base.py').write_text("model = dict(type='X', backbone=dict(depth=2, width=8), head=dict(a=1))\n"
                                      "total_epochs = 3\nlog_config = dict(interval=10)\n")
    (tmp_path / 'child.py').write_text("_base_ = ['base.py']\nvideos_per_gpu = 4\n"
                                       "model = dict(backbone=dict(depth=4), head=dict(_delete_=True, b=2))\n"
                                       "optimizer = dict(type='AdamW', base_lr=1e-4, weight_decay=0.1)\n")
    cfg = Config.fromfile(str(tmp_path / 'child.py'))
    assert cfg.model.type == 'X' and cfg.model.backbone == dict(depth=4, width=8)      # recursive merge
    assert cfg.model.head == dict(b=2)                                                # _delete_ replaces
    assert cfg.total_epochs == 3 and cfg.log_config.interval == 10
    cfg.merge_from_dict(parse_cfg_options(['model.backbone.width=16', 'total_epochs=1', 'tag=run7', 'x.y=[1,2]']))
    assert cfg.model.backbone.width == 16 and cfg.total_epochs == 1 and cfg.tag == 'run7' and cfg.x.y == [1, 2]
    # linear scaling rule, tools/train.py:160-166: base_lr is popped, lr = base_lr * videos_per_gpu * world_size
    assert abs(scaled_lr(cfg, 8) - 1e-4 * 4 * 8) < 1e-12 and 'base_lr' not in cfg.optimizer
    assert scaled_lr(cfg, 2) == cfg.optimizer['lr']                                   # second call: nothing to scale


class _FakeStepper(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(1))
        self.seen = []

    def train_step(self, batch, optimizer):
        self.seen.append(batch)
        return dict(loss=torch.tensor(float(len(self.seen))), log_vars=dict(loss=float(len(self.seen))), num_samples=1)


def test_runner_multi_loader_interleave_and_hooks(tmp_path):
    """clover_runner.py:60-96: one step per loader per batch index; once the shorter loader runs out it is
    restarted and from then on EVERY slot of a row draws from the restarted iterator (the longer loader's own batch
    of that row is dropped) — reproduced, it decides which samples the reference trains on."""
    from clover_amd.runner import CheckpointHook, CloverRunner, Hook, LogHook
    A, B = [f'a{i}' for i in range(5)], [f'b{i}' for i in range(3)]
    calls = []

    class Spy(Hook):
        def before_run(self, r): calls.append('before_run')
        def before_train_epoch(self, r): calls.append('before_epoch')
        def before_train_iter(self, r): calls.append('bi')
        def after_train_iter(self, r): calls.append('ai')
        def after_train_epoch(self, r): calls.append('after_epoch')
        def after_run(self, r): calls.append('after_run')
    m = _FakeStepper()
    r = CloverRunner(m, max_epochs=1, work_dir=str(tmp_path))
    log = LogHook(interval=1)
    for h in (Spy(), log, CheckpointHook(str(tmp_path))):
        r.register_hook(h)
    r.run([A, B], [('train', 1)], 1)
    assert m.seen == ['a0', 'b0', 'a1', 'b1', 'a2', 'b2', 'a3', 'b0', 'b1', 'b2']
    assert r.iter == 5 and r.epoch == 1                      # :91 counts batch indices
    assert calls == ['before_run', 'before_epoch'] + ['bi', 'ai'] * 10 + ['after_epoch', 'after_run']
    assert [rec['loss'] for rec in log.records] == [float(i) for i in range(1, 11)]
    # single loader mode (:17-35) and the checkpoint layout {'meta','state_dict','optimizer'?}
    m2 = _FakeStepper()
    r2 = CloverRunner(m2, max_epochs=2, work_dir=str(tmp_path))
    r2.run([A], [('train', 1)])
    assert m2.seen == A + A and r2.iter == 10 and r2.epoch == 2
    ck = torch.load(str(tmp_path / 'epoch_1.pth'))
    assert set(ck) >= {'meta', 'state_dict'} and ck['meta']['epoch'] == 1 and 'w' in ck['state_dict']
    r3 = CloverRunner(_FakeStepper(), max_epochs=3)
    r3.resume(str(tmp_path / 'epoch_1.pth'))
    assert r3.epoch == 1 and r3.iter == 5


def test_runner_short_loader_runs_dry_like_the_reference():
    """With too short a second loader the restarted iterator is exhausted mid-epoch and `next()` raises — the
    reference has no guard (clover_runner.py:80-82) and neither has the drop-in."""
    from clover_amd.runner import CloverRunner
    r = CloverRunner(_FakeStepper(), max_epochs=1)
    with pytest.raises(StopIteration):
        r.run([['a0', 'a1', 'a2'], ['b0']], [('train', 1)])


# ----------------------------------------------------------------------------- retrieval metrics
def test_recall_for_video_text_retrieval_goldens():
    """clover_amd.evaluation against the reference's own metric values (g_finetune.npz; accuracy.py:430-462)."""
    from clover_amd.evaluation import normalize_fn, recall_for_video_text_retrieval
    g = gutil.load('g_finetune.npz')
    keys = ['Recall@1', 'Recall@5', 'Recall@10', 'MR', 'Recall@all']
    for N, D in ((1, 8), (7, 16), (50, 32), (200, 64)):
        ve = cf.cf_float(f'recall.N{N}.v', (N, D), 1.0).numpy()
        te = (0.35 * ve + cf.cf_float(f'recall.N{N}.t', (N, D), 1.0).numpy()).astype(np.float32)
        if N == 7:
            te[3] = 0
        m = recall_for_video_text_retrieval(ve, te)
        np.testing.assert_allclose([m[k] for k in keys], g[f'recall.N{N}'], rtol=0, atol=1e-9)
        mt = recall_for_video_text_retrieval(torch.from_numpy(ve), torch.from_numpy(te))      # tensors accepted
        assert mt == m
    sc = cf.cf_float('recall.scores', (12, 12), 1.0).numpy()
    m = recall_for_video_text_retrieval(input_scores=sc)
    np.testing.assert_allclose([m[k] for k in keys], g['recall.scores'], rtol=0, atol=1e-9)
    z = normalize_fn(np.array([[3.0, 4.0], [0.0, 0.0]]))
    assert np.array_equal(z, np.array([[0.6, 0.8], [0.0, 0.0]]))


def test_finetune_registered_and_refuses_without_gpu():
    """CloverFinetune / NormSoftmaxLoss are registry entries with the reference's constructor kwargs; the loss has
    no CPU path."""
    import clover_amd
    from clover_amd.builder import LOSSES, RECOGNIZERS
    assert 'CloverFinetune' in RECOGNIZERS.module_dict and 'NormSoftmaxLoss' in LOSSES.module_dict
    m = clover_amd.build_model(cf.tiny_finetune_cfg())
    assert m.task == 'retrieval' and m.loss_func.use_cos_similarity and m.loss_func.t == 0.05
    assert not any(k.startswith(('mlm_head', 'mlm_ssl')) for k in m.state_dict())
    with pytest.raises(RuntimeError):
        m.loss_func(torch.randn(4, 8), torch.randn(4, 8))


def test_tools_cli_host_side():
    """tools/train.py and tools/test.py: the reference's argument names parse, and the synthetic test loader shards the
    test set rank-major without losing or duplicating a pair (no GPU involved)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mods = {}
    for name in ('train', 'test'):
        spec = importlib.util.spec_from_file_location(f'clv_tools_{name}', os.path.join(root, 'tools', f'{name}.py'))
        mods[name] = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mods[name])
    argv = sys.argv
    try:
        sys.argv = ['train.py', 'cfg.py', '--work_dir', 'w', '--resume-from', 'a.pth', '--seed', '3', '--launcher', 'none',
                    '--cfg-options', 'total_epochs=1', 'optimizer.base_lr=0.1']
        a = mods['train'].parse_args()
        assert (a.config, a.work_dir, a.resume_from, a.seed, a.launcher) == ('cfg.py', 'w', 'a.pth', 3, 'none')
        assert a.cfg_options == ['total_epochs=1', 'optimizer.base_lr=0.1']
        sys.argv = ['test.py', 'cfg.py', 'ckpt.pth', '--eval', 'recall_for_video_text_retrieval', '--gpu-collect',
                    '--out', 'o.json']
        t = mods['test'].parse_args()
        assert (t.config, t.checkpoint, t.eval, t.out) == ('cfg.py', 'ckpt.pth', ['recall_for_video_text_retrieval'], 'o.json')
    finally:
        sys.argv = argv
    seen = []
    for rank in range(3):
        ld = mods['test'].SyntheticTestLoader(pairs=10, batch=2, frames=2, tokens=8, rank=rank, world=3, device='cpu')
        for b in ld:
            assert b['imgs'].shape[0] == b['index'].numel() == b['token_ids'].shape[0]
            seen += b['index'].tolist()
    assert sorted(seen) == list(range(10))


def test_grouped_weight_gradient_chunking():
    """ops.wgrad_chunks: <= 80 problems per launch, and two in-place problems (few rows, or a large 256-divisible output)
    with the same dW never in one launch — the in-place kernels add without atomics."""
    import torch
    from clover_amd import ops, _lib
    L = _lib.lib()
    assert L.clv_linear_wgrad_in_place(512, 768, 3072) == 1 and L.clv_linear_wgrad_in_place(3648, 768, 3072) == 1
    assert L.clv_linear_wgrad_in_place(12544, 384, 1536) == 0 and L.clv_linear_wgrad_in_place(3136, 768, 768) == 0
    assert L.clv_linear_wgrad_class(12544, 384, 384) == 0 and L.clv_linear_wgrad_class(3136, 768, 768) == 1
    assert L.clv_linear_wgrad_class(512, 768, 768) == 0                       # few-row problems stay on 128 x 128 tiles
    shared = torch.zeros(768, 768)
    other = [torch.zeros(768, 768) for _ in range(3)]
    item = lambda dw, M: (None, None, dw, None, M, 768, 768)
    # partial-mode uses of a shared dW may sit together in the GEMM launch (each writes its own partial buffer; their
    # folds are separated below) ...
    assert [len(c) for c in ops.wgrad_chunks([item(shared, 12544), item(shared, 12544), item(other[0], 512)])] == [3]
    # ... two in-place uses may not; an in-place + a partial-mode use may
    assert [len(c) for c in ops.wgrad_chunks([item(shared, 512), item(other[0], 512), item(shared, 640),
                                              item(shared, 12544), item(other[1], 512)])] == [2, 3]
    many = [item(torch.zeros(8, 8), 12544) for _ in range(175)]
    assert [len(c) for c in ops.wgrad_chunks(many)] == [80, 80, 15]
    # the folds of two partial-mode uses of ONE dW (or db) never share a launch: the fold kernel adds without atomics
    w = [torch.zeros(4) for _ in range(4)]
    fold = lambda dw, db=None: (torch.zeros(1), dw, db, 2, 2, 4)
    assert [len(c) for c in ops.fold_chunks([fold(shared), fold(other[0]), fold(shared), fold(other[1])])] == [2, 2]
    assert [len(c) for c in ops.fold_chunks([fold(other[0], w[0]), fold(other[1], w[0]), fold(other[2], w[1])])] == [1, 2]
    assert [len(c) for c in ops.fold_chunks([fold(torch.zeros(2, 2)) for _ in range(130)])] == [64, 64, 2]


def test_fusion_text_outputs_and_token_mean_equal_autograd():
    """ops.fusion_text_outputs (one autograd node for the MLM decoder's input + the two reconstruction heads' CLS rows,
    multimodal_transformer_pretrain.py:129,148-149,156-157) and ops.token_mean (the vision head's spatial average,
    ssl_head.py:88-94) against the plain slice / unbind / select / mean expressions they replace: values and gradients, with
    both consumers, with either one alone (the ablation switches), in fp32 and bf16."""
    import torch
    from clover_amd import ops
    torch.manual_seed(3)
    B, S, L, D = 3, 11, 4, 8
    for dtype in (torch.float32, torch.bfloat16):
        h0 = torch.randn(2 * B, S, D).to(dtype)
        wt, wc = torch.randn(B, L, D), torch.randn(2, B, D)
        for use_t, use_c in ((True, True), (True, False), (False, True)):
            h = h0.clone().requires_grad_()
            t_last, cls = ops.fusion_text_outputs(h, S - L)
            assert t_last.is_contiguous() and t_last.dtype == dtype and cls.dtype == torch.float32
            loss = (t_last.float() * wt).sum() * use_t + (cls * wc).sum() * use_c
            loss.backward()
            hr = h0.clone().requires_grad_()
            t_all = hr[:, S - L:]
            tl, vf = t_all.unflatten(0, (2, B)).unbind(0)
            cr = torch.stack([tl[:, 0].float(), vf[:, 0].float()])
            lr = (tl.float() * wt).sum() * use_t + (cr * wc).sum() * use_c
            lr.backward()
            assert torch.equal(t_last, tl) and torch.equal(cls, cr)
            tol = 0.0 if dtype == torch.float32 else 2e-2       # (bf16: the CLS rows are added in fp32 here, in bf16 there)
            assert (h.grad.float() - hr.grad.float()).abs().max().item() <= tol * hr.grad.float().abs().max().item() + 1e-6
    for dtype in (torch.float32, torch.bfloat16):
        x0 = torch.randn(4, 2, 3, 3, D).to(dtype)
        w = torch.randn(4, D)
        x = x0.clone().requires_grad_()
        y = ops.token_mean(x)
        ((y * w).sum() + (x.float() ** 2).sum()).backward()     # a second consumer, as the fusion encoder is
        xr = x0.clone().requires_grad_()
        yr = xr.float().mean(dim=(1, 2, 3))
        ((yr * w).sum() + (xr.float() ** 2).sum()).backward()
        assert y.dtype == torch.float32 and (y - yr).abs().max().item() < 1e-6
        assert (x.grad.float() - xr.grad.float()).abs().max().item() <= 2e-2 * xr.grad.float().abs().max().item()


def test_sumsq_range_table_host_logic():
    """ops.sumsq_range_table (the block table of clv_sumsq_ranges): ranges are cut into blocks of <= CLV_SUMSQ_CHUNK floats,
    nothing is lost or counted twice, an empty list gives zero blocks."""
    from clover_amd import ops
    hdr = open(os.path.join(ROOT, 'include', 'clover_hip.h')).read()
    assert ops.SUMSQ_CHUNK == int(re.search(r'#define CLV_SUMSQ_CHUNK (\d+)', hdr).group(1))
    assert ops.SUMSQ_SLOTS == int(re.search(r'#define CLV_SUMSQ_SLOTS (\d+)', hdr).group(1))
    ranges = [(0, 7), (8, 8), (16, 40000), (40004, 40005), (100000, 300000)]
    tab, n = ops.sumsq_range_table(ranges, 'cpu')
    assert tab.shape == (n, 2) and int(tab[:, 1].max()) <= ops.SUMSQ_CHUNK
    covered = torch.zeros(300000, dtype=torch.int32)
    for off, cnt in tab.tolist():
        assert off % 4 == 0
        covered[off:off + cnt] += 1
    want = torch.zeros_like(covered)
    for a, b in ranges:
        want[a:b] = 1
    assert torch.equal(covered, want)
    tab0, n0 = ops.sumsq_range_table([], 'cpu')
    assert n0 == 0
