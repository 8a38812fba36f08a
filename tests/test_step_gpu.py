"""GPU parity of the registered modules and of the whole pre-training step against
(a) the goldens produced by the reference itself and (b) the oracle, on the same closed-form
weights/inputs.  BASELINE config 1 (tiny 2-stage Swin + BERT-tiny), a mid-width model whose Linear layers all run on the
HIP GEMMs, and BASELINE configs 2 / 4 / 5 at full size.  `-m gpu` only.

Tolerances (DESIGN.md "Parity"): indexing/masking bit-exact.  The fp32 kernels (focal CE, exclusive
InfoNCE + rank) are checked at 1e-4 .. 1e-3.  The north-star bound on the step losses (1e-3) is asserted in
tests/test_parity_gpu.py on the fp32 parity mode of the same path (measured 1e-6 .. 8e-6).  THIS file runs the 16-bit
training path: every activation is stored in 16 bits and every GEMM operand is 16-bit.  LOSS_TOL below is, per key, 1.5 x
the largest |loss - reference| the path has been MEASURED at; feature maps / gradients are relative to their max magnitude.

Two builds of the same kernels (clover_amd/_lib.py, csrc/common.hpp):

* fp16 (default since round 6: the reference's own arithmetic type, configs/exp_local/pretrain_webvid_cc3m.py:21) — three
  more significand bits than bf16 at the same MFMA rate and the same step time.  Measured |loss - reference| (round 6):

    workload (reference)                       mlm      nce      rank_t_tm  v_nce    rank_v_vm  total
    config 1, B = 1 (golden)                   1.7e-4   0        2.2e-3     0        5.4e-4     2.6e-3
    config 1, B = 2 (golden)                   5.1e-5   1.1e-3   4.1e-4     1.6e-3   1.0e-5     2.3e-3
    config 1, B = 4 (golden)                   5.8e-5   3.9e-5   1.2e-3     9.4e-4   5.5e-4     3.1e-4
    mid widths 96 / 192, B = 2 (golden)        1.4e-4   1.0e-3   7.9e-4     1.6e-3   8.2e-4     8.6e-4
    mid widths 96 / 192, B = 4 (golden)        1.7e-4   3.3e-4   4.1e-4     5.8e-4   1.6e-3     9.4e-4
    Swin-T 8 f full size, B = 2 (oracle)       2.8e-4   5.3e-4   8.6e-4     1.8e-3   2.5e-4     2.1e-3
    Swin-B 16 f full size, B = 2 (oracle)      2.5e-5   3.1e-4   7.2e-4     9.6e-4   1.1e-4     2.1e-3
    Swin-B 32 f full size, B = 2 (oracle)      1.9e-6   1.7e-4   8.3e-4     1.3e-4   5.1e-4     2.8e-4
    Swin-T 8 f, B = 8 = the BENCH shapes,
      engine + hipGraphs (reference golden)    1.2e-5   8.6e-5   1.0e-3     1.6e-3   2.7e-4     3.0e-3
    Swin-T 16 f / 32 f, B = 2 (oracle)         3.2e-4   3.7e-4   1.8e-3     4.6e-4   6.9e-4     1.1e-3
    max                                        3.2e-4   1.1e-3   2.2e-3     1.8e-3   1.6e-3     3.0e-3
    LOSS_TOL_F16 (~1.5 x; mlm rounded up)      1.0e-3   2.0e-3   3.3e-3     2.8e-3   2.4e-3     4.5e-3

* bf16 (CLOVER_HALF=bf16; tests/test_bf16_build_gpu.py runs this file once more in a child process).  The isolation runs
  of tools/parity_isolate.py (profiles/r03_parity_isolate.json) show that bf16 activation storage or bf16 GEMM operands alone
  already move the contrastive / rank losses by 2e-3 .. 1e-2 (cosine logits are divided by the temperature 0.05).  Measured
  (round 5 run; "all rounds" adds the maxima rounds 3-4 recorded, bench line included — the contrastive terms move by a few
  1e-3 between boxes / builds because the loss kernels accumulate atomically):

    workload (reference)                       mlm      nce      rank_t_tm  v_nce    rank_v_vm  total
    config 1, B = 1 (golden)                   3.0e-3   0        6.7e-3     0        9.5e-3     1.3e-2
    config 1, B = 2 (golden)                   1.1e-3   3.6e-3   8.4e-3     1.1e-2   1.0e-2     3.4e-3
    config 1, B = 4 (golden)                   5.8e-4   2.0e-3   8.9e-4     5.7e-3   7.1e-3     2.5e-4
    Swin-T 8 f full size, B = 2 (oracle)       8.9e-5   4.4e-3   4.7e-3     7.5e-3   1.3e-3     9.2e-3
    Swin-B 16 f full size, B = 2 (oracle)      1.5e-3   1.7e-3   3.6e-3     1.5e-2   4.8e-4     1.8e-2
    Swin-B 32 f full size, B = 2 (oracle)      2.0e-3   2.2e-3   6.7e-3     1.9e-3   1.9e-3     1.0e-2
    round 6 (strict own GEMMs everywhere below; the three full-size rows above re-measured: 8.7e-5 .. 1.7e-2, unchanged):
    Swin-T 8 f, B = 8 = the BENCH shapes,
      engine + hipGraphs (reference golden)    1.9e-4   5.8e-3   4.1e-3     9.7e-4   6.1e-3     5.1e-3
    mid widths 96 / 192, B = 2 (golden)        8.5e-4   5.9e-3   2.0e-3     1.4e-2   5.4e-4     6.1e-3
    mid widths 96 / 192, B = 4 (golden)        1.4e-3   2.1e-3   2.9e-3     1.4e-2   4.5e-3     1.6e-2
    config 1, B = 1 (golden), few-row GEMMs on
      the HIP kernels (were library below 64)  5.4e-3   0        ...
    max, all rounds                            5.4e-3   9.6e-3   9.6e-3     1.64e-2  1.1e-2     1.84e-2
    LOSS_TOL_BF16 = 1.5 x that                 8.1e-3   1.5e-2   1.5e-2     2.5e-2   1.7e-2     2.8e-2
"""
import numpy as np
import pytest
import torch

import closed_form as cf
import gutil

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LOSS_KEYS = ['mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss']
# per key, 1.5 x the largest |loss - reference| measured (tables in the module docstring): the bf16 build of the kernels
# (CLOVER_HALF=bf16) and the default fp16 build (three more significand bits: the reference's own arithmetic type)
LOSS_TOL_BF16 = dict(mlm_loss=8.1e-3, nce_loss=1.5e-2, rank_t_tm_loss=1.5e-2, v_nce_loss=2.5e-2, rank_v_vm_loss=1.7e-2, loss=2.8e-2)
LOSS_TOL_F16 = dict(mlm_loss=1e-3, nce_loss=2e-3, rank_t_tm_loss=3.3e-3, v_nce_loss=2.8e-3, rank_v_vm_loss=2.4e-3, loss=4.5e-3)
from clover_amd import _lib as _clv_lib  # noqa: E402
LOSS_TOL = LOSS_TOL_F16 if _clv_lib.HALF_F16 else LOSS_TOL_BF16


def grad_tol(name):
    """max|err| / max|ref| bound for a parameter gradient.  Parameters whose ONLY gradient source is a
    contrastive loss (the projector / masked-feature heads feeding cosine logits divided by the
    temperature 0.05) inherit that x20 amplification of the bf16 backbone rounding: 1e-1.  Everything
    else (backbones, fusion, MLM head; observed <= 2e-2): 3e-2."""
    return 1e-1 if name.startswith(('ssl_head.', 'mlm_ssl_V_head.', 'mlm_ssl_T_head.')) else 3e-2


def rel(a, b):
    a = a.detach().float().cpu().numpy().astype(np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-20)


def rel_packed(g, name, t):
    sub, stats = gutil.packed(t)
    gsub = g[name + '.sub'].astype(np.float64)
    assert g[name + '.stats'][2] == stats[2]
    return np.abs(sub - gsub).max() / max(np.abs(gsub).max(), 1e-20)


@pytest.fixture(scope='module')
def model():
    import clover_amd
    m = clover_amd.build_model(cf.tiny_model_cfg())
    sd = cf.cf_state(gutil.manifest())
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert all('relative_position_index' in k for k in missing) and not unexpected
    return m.to(DEV).eval()


def to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


def test_swin_backbone(model):
    g = gutil.load('g_swin.npz')
    b = cf.cf_batch(2, tag='swin')
    x, vm = b['imgs'][:, 0].to(DEV), b['v_token_mask'].to(DEV)
    with torch.no_grad():
        y = model.backbone(x)
        ym, w = model.backbone(x.clone(), vm)
        yc2, ym2 = model.backbone.forward_pair(x, vm)
        yb = model.backbone.forward_both(x, vm)
    assert y.shape == g['clean.out'].shape
    assert rel(y, g['clean.out']) < 3e-2, rel(y, g['clean.out'])
    assert rel(ym, g['masked.out']) < 3e-2
    assert np.array_equal(w.float().cpu().numpy().astype(np.int8), g["masked.w"])          # bit-exact blend map
    # the batched 2B pass equals the two separate passes
    assert torch.equal(yc2.permute(0, 4, 1, 2, 3), y) and torch.equal(ym2.permute(0, 4, 1, 2, 3), ym)
    assert torch.equal(yb, torch.cat([yc2, ym2], 0))           # stacked patch-embed output, no cat


def test_text_and_fusion(model):
    g = gutil.load('g_bert_fuse.npz')
    b = cf.cf_batch(3, tag='bf')
    ids, mask = b['token_ids'][:, 0].to(DEV), b['input_mask'][:, 0].to(DEV)
    with torch.no_grad():
        t = model.text_backbone(ids, mask)['last_hidden_state']
        assert rel(t, g['bert.last_hidden_state']) < 2e-2
        vt = cf.cf_float('bf.vt', (3, 2, 196, 96), 1.0).to(DEV)
        tt = torch.from_numpy(g['bert.last_hidden_state']).to(DEV)
        f = model.multimodal_backbone(visual_token=vt, text_input_mask=mask, text_input_embeds=tt)
    assert rel(f['t_last_hidden_state'], g['fuse.t_last_hidden_state']) < 2e-2
    assert rel_packed(g, 'fuse.v_last_hidden_state', f['v_last_hidden_state']) < 3e-2


def test_heads(model):
    g = gutil.load('g_heads_loss.npz')
    vis = cf.cf_float('hl.vis', (4, 96, 2, 14, 14), 1.0).to(DEV)
    txt = cf.cf_float('hl.txt', (4, 16, 128), 1.0).to(DEV)
    row = cf.cf_float('hl.row', (4, 128), 1.0).to(DEV)
    with torch.no_grad():
        assert rel(model.ssl_head.forward_vision(vis), g['mm.vision']) < 2e-2
        assert rel(model.ssl_head.forward_vision(vis[:1]), g['mm.vision_b1']) < 2e-2
        assert rel(model.ssl_head.forward_text(txt), g['mm.text']) < 2e-2
        assert rel(model.mlm_ssl_V_head(row), g['V.head']) < 2e-2
        assert rel(model.mlm_ssl_T_head(row), g['T.head']) < 2e-2
        assert rel(model.mlm_head(txt[:2]), g['mlm.scores']) < 2e-2
        logits = cf.cf_float('hl.logits', (7, 1024), 4.0).to(DEV)
        tgt = cf.cf_int('hl.tgt', (7,), 0, 1024).to(DEV)
        assert abs(model.mlm_loss_func(logits, tgt).item() - float(g['focal'])) < 1e-4
        for G in [1, 2, 4, 8]:
            e = [cf.cf_float(f'hl.e{k}.{G}', (G, 128), 1.0).to(DEV) for k in range(4)]
            l = model.ssl_loss(*e)
            assert abs(l['nce_loss'].item() - float(g[f'nce.G{G}.nce_loss'])) < 1e-3
            assert abs(l['rank_t_tm_loss'].item() - float(g[f'nce.G{G}.rank_t_tm_loss'])) < 1e-3


@pytest.mark.parametrize('B', [1, 3])
def test_forward_test_embeddings(model, B):
    """``forward_test(separate_test=True)`` (multimodal_transformer_pretrain.py:194-218) — one Swin
    pass + one BERT pass + the two projection heads — against the reference's own embeddings."""
    g = gutil.load('g_test.npz')
    batch = to_dev(cf.cf_batch(B, tag=f'test{B}'))
    was_training = model.training
    model.eval()
    try:
        with torch.no_grad():
            v, t = model.forward_test(batch['imgs'], token_ids=batch['token_ids'], segment_ids=batch['segment_ids'],
                                      input_mask=batch['input_mask'])
            v2, t2 = model(batch['imgs'], None, return_loss=False, token_ids=batch['token_ids'],
                           segment_ids=batch['segment_ids'], input_mask=batch['input_mask'])
    finally:
        model.train(was_training)
    assert v.shape == (B, 128) and t.shape == (B, 128) and v.dtype == torch.float32
    assert rel(v, g[f'B{B}.visual_emb']) < 2e-2, rel(v, g[f'B{B}.visual_emb'])
    assert rel(t, g[f'B{B}.text_emb']) < 2e-2, rel(t, g[f'B{B}.text_emb'])
    assert torch.equal(v, v2) and torch.equal(t, t2)             # BaseRecognizer.forward(return_loss=False) route


@pytest.mark.parametrize('B', [1, 2, 4])
def test_step_losses_and_grads(model, B):
    """BASELINE config 1: full train_step; losses vs the reference's own numbers."""
    g = gutil.load('g_step.npz')
    batch = to_dev(cf.cf_batch(B, tag=f'step{B}'))
    model.zero_grad(set_to_none=True)
    out = model.train_step(batch, None)
    assert out['num_samples'] == B
    lv = out['log_vars']
    errs = {k: abs(lv[k] - float(g[f'B{B}.{k}'])) for k in LOSS_KEYS}
    print('loss errors', B, errs)
    for k in LOSS_KEYS:
        assert errs[k] <= LOSS_TOL[k], (k, lv[k], float(g[f'B{B}.{k}']))
    out['loss'].backward()
    named = dict(model.named_parameters())
    worst = {}
    for k in [n[len(f'B{B}.grad.'):-4] for n in g.files if n.startswith(f'B{B}.grad.') and n.endswith('.sub')]:
        worst[k] = rel_packed(g, f'B{B}.grad.{k}', named[k].grad)
    print('grad rel errors', B, worst)
    for k, e in worst.items():
        assert e < grad_tol(k), (k, e)
    unused = sorted(k for k, p in named.items() if p.grad is None)
    assert unused == gutil.unused_params()


@pytest.mark.parametrize('switch', ['symmetry_rank', 'mlm_ssl_head', 'mlm_head', 'mlm_loss'])
def test_forward_train_ablation_switches(model, switch):
    """The constructor switches of CloverPretrain.forward_train (multimodal_transformer_pretrain.py:129-169): with
    symmetry_rank=False the second ssl_loss call disappears, with mlm_ssl_head=None (and symmetry_rank=False) both do, with
    mlm_head=None the MLM loss; the losses that remain keep the value they have in the full recipe (reference golden) and
    equal the oracle run with the same switch; no gradient reaches a switched-off head.  mlm_loss=None routes the MLM
    scores through loss_type's CrossEntropyLoss over the labelled rows (:141-142)."""
    import clover_amd
    from oracle import model as om
    cfg = cf.tiny_model_cfg()
    ocfg = dict(cf.oracle_cfg_from(cf.tiny_model_cfg()))
    if switch == 'symmetry_rank':
        cfg['symmetry_rank'] = False
        ocfg['symmetry_rank'] = False
        gone = {'v_nce_loss', 'rank_v_vm_loss'}
    elif switch == 'mlm_ssl_head':
        cfg['symmetry_rank'], cfg['mlm_ssl_head'] = False, None
        ocfg['symmetry_rank'], ocfg['mlm_ssl_head'] = False, False
        gone = {'v_nce_loss', 'rank_v_vm_loss', 'nce_loss', 'rank_t_tm_loss'}
    elif switch == 'mlm_head':
        cfg['mlm_head'] = None
        ocfg['mlm_head'] = False
        gone = {'mlm_loss'}
    else:
        cfg['mlm_loss'] = None
        gone = set()
    m = clover_amd.build_model(cfg)
    sd = cf.cf_state(gutil.manifest())
    missing, unexpected = m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()}, strict=False)
    assert not unexpected and all('relative_position_index' in k for k in missing), (missing, unexpected)
    m = m.to(DEV).eval()
    B = 2
    batch_cpu = cf.cf_batch(B, tag=f'step{B}')
    out = m.train_step(to_dev(batch_cpu), None)
    lv = out['log_vars']
    g = gutil.load('g_step.npz')
    assert set(lv) == (set(LOSS_KEYS) - gone), (set(lv), gone)
    if switch != 'mlm_loss':
        ref = om.forward_train(sd, batch_cpu, ocfg, gather=False)
        assert set(ref) == set(lv) - {'loss'}
        for k in set(lv) - {'loss'}:
            assert abs(lv[k] - float(ref[k])) <= LOSS_TOL[k], (k, lv[k], float(ref[k]))
            assert abs(lv[k] - float(g[f'B{B}.{k}'])) <= LOSS_TOL[k], (k, lv[k], float(g[f'B{B}.{k}']))
    else:
        # CrossEntropyLoss (mean over the labelled rows) of the same scores the focal loss reads in the full recipe
        full = model.train_step(to_dev(batch_cpu), None)['log_vars']
        for k in ('nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss'):
            assert abs(lv[k] - full[k]) <= 1e-3 * max(1.0, abs(full[k])), k
        assert lv['mlm_loss'] > full['mlm_loss'] > 0            # focal (gamma = 2) down-weights the same cross entropy
    out['loss'].backward()
    named = dict(m.named_parameters())
    if switch in ('symmetry_rank', 'mlm_ssl_head'):
        dead = [n for n in named if n.startswith('mlm_ssl_T_head.')]
        assert all(named[n].grad is None or float(named[n].grad.abs().max()) == 0.0 for n in dead)
    assert all(torch.isfinite(p.grad).all() for p in named.values() if p.grad is not None)


# ----------------------------------------------------------------------------- the HIP GEMMs against reference goldens
@pytest.fixture(scope='module')
def mid_model():
    """cf.mid_model_cfg: VideoSwin-T's stage-0 / 1 widths (96 / 192) + BERT-tiny — every Linear on the HIP GEMM kernels."""
    import clover_amd
    m = clover_amd.build_model(cf.mid_model_cfg())
    missing, unexpected = m.load_state_dict(cf.cf_state(gutil.manifest('mid')), strict=False)
    assert all('relative_position_index' in k for k in missing) and not unexpected
    return m.to(DEV).eval()


@pytest.mark.usefixtures('strict_own_gemm')
def test_mid_modules_match_reference(mid_model):
    """Reference-GENERATED activations at widths the HIP GEMMs take (config 1's 48 / 96 widths run their
    Linear layers on the library): per-block Swin outputs (stage 0 on the fused row-streaming kernels, stage 1 on
    clv_gemm_nt, the merge), the text tower and the fusion encoder — no library GEMM anywhere (strict_own_gemm)."""
    g = gutil.load('g_mid.npz')
    b = cf.cf_batch(2, tag='mid.swin')
    x, vm = b['imgs'][:, 0].to(DEV), b['v_token_mask'].to(DEV)
    bb = mid_model.backbone
    taps = {}
    hooks = []
    for i, layer in enumerate(bb.layers):
        for j, blk in enumerate(layer.blocks):
            hooks.append(blk.register_forward_hook(lambda mod, a, o, k=f'layers.{i}.blocks.{j}': taps.__setitem__(k, o)))
    with torch.no_grad():
        y = bb(x)
        ym, _ = bb(x.clone(), vm)
    for h in hooks:
        h.remove()
    assert rel_packed(g, 'swin.clean.out', y) < 3e-2 and rel_packed(g, 'swin.masked.out', ym) < 3e-2
    b3 = cf.cf_batch(3, tag='mid.bf')
    ids, mask = b3['token_ids'][:, 0].to(DEV), b3['input_mask'][:, 0].to(DEV)
    with torch.no_grad():
        t = mid_model.text_backbone(ids, mask)['last_hidden_state']
        assert rel(t, g['bert.last_hidden_state']) < 2e-2
        vt = cf.cf_float('mid.bf.vt', (3, 2, 196, 192), 1.0).to(DEV)
        tt = torch.from_numpy(g['bert.last_hidden_state']).to(DEV)
        f = mid_model.multimodal_backbone(visual_token=vt, text_input_mask=mask, text_input_embeds=tt)
    assert rel(f['t_last_hidden_state'], g['fuse.t_last_hidden_state']) < 2e-2
    assert rel_packed(g, 'fuse.v_last_hidden_state', f['v_last_hidden_state']) < 3e-2


@pytest.mark.usefixtures('strict_own_gemm')
@pytest.mark.parametrize('B', [2, 4])
def test_mid_step_losses_and_grads(mid_model, B):
    """The full train_step at the mid widths: six losses and 23 parameter gradients (every Linear family: fused stage-0
    qkv / proj / fc1 / fc2, stage-1 GEMMs, merge, text tower, fc_in, fusion encoder, MLM transform / decoder, heads) against
    the REFERENCE's own numbers (g_mid.npz), forward / input-gradient / weight-gradient GEMMs all on the HIP kernels."""
    g = gutil.load('g_mid.npz')
    batch = to_dev(cf.cf_batch(B, tag=f'mid.step{B}'))
    mid_model.zero_grad(set_to_none=True)
    out = mid_model.train_step(batch, None)
    lv = out['log_vars']
    errs = {k: abs(lv[k] - float(g[f'B{B}.{k}'])) for k in LOSS_KEYS}
    print('mid loss errors', B, errs)
    for k in LOSS_KEYS:
        assert errs[k] <= LOSS_TOL[k], (k, lv[k], float(g[f'B{B}.{k}']))
    out['loss'].backward()
    named = dict(mid_model.named_parameters())
    worst = {}
    for k in [n[len(f'B{B}.grad.'):-4] for n in g.files if n.startswith(f'B{B}.grad.') and n.endswith('.sub')]:
        worst[k] = rel_packed(g, f'B{B}.grad.{k}', named[k].grad)
    print('mid grad rel errors', B, worst)
    for k, e in worst.items():
        assert e < grad_tol(k), (k, e)
    assert sum(p.grad is None for p in named.values()) == int(g[f'B{B}.n_unused'])
    mid_model.zero_grad(set_to_none=True)


# ----------------------------------------------------------------------------- retrieval fine-tuning
@pytest.fixture(scope='module')
def ft_model():
    import clover_amd
    m = clover_amd.build_model(cf.tiny_finetune_cfg())
    sd = {k: v for k, v in cf.cf_state(gutil.manifest()).items() if k in m.state_dict()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert all('relative_position_index' in k for k in missing) and not unexpected
    return m.to(DEV).eval()


FT_AUX = ['token_ids', 'segment_ids', 'input_mask']


@pytest.mark.parametrize('B', [2, 4])
def test_finetune_retrieval_step(ft_model, B):
    """CloverFinetune(task='retrieval').train_step (multimodal_transformer_finetune.py:59-86) against the
    reference's own loss and gradients; the oracle on the same inputs as a second witness."""
    from oracle import model as om
    g = gutil.load('g_finetune.npz')
    batch = to_dev(cf.cf_batch(B, tag=f'ft{B}'))
    ft_model.zero_grad(set_to_none=True)
    out = ft_model.train_step({k: batch[k] for k in ['imgs', 'label'] + FT_AUX}, None)
    assert out['num_samples'] == B
    lv = out['log_vars']
    assert set(lv) == {'retrieval_nce_loss', 'loss'}
    ref = float(g[f'train.B{B}.loss'])
    assert abs(lv['loss'] - ref) <= 3e-2, (lv['loss'], ref)            # cosine logits / 0.05: the nce tolerance
    P = cf.cf_state(gutil.manifest())
    o = om.finetune_forward_train(P, cf.cf_batch(B, tag=f'ft{B}'), cf.oracle_cfg_from(cf.tiny_model_cfg()),
                                  gather=False)['retrieval_nce_loss'].item()
    assert abs(lv['loss'] - o) <= 3e-2, (lv['loss'], o)
    out['loss'].backward()
    named = dict(ft_model.named_parameters())
    worst = {}
    for k in [n[len(f'train.B{B}.grad.'):-4] for n in g.files if n.startswith(f'train.B{B}.grad.') and n.endswith('.sub')]:
        worst[k] = rel_packed(g, f'train.B{B}.grad.{k}', named[k].grad)
    print('finetune grad rel errors', B, worst)
    for k, e in worst.items():
        assert e < 1e-1, (k, e)                                       # every gradient here is contrastive-only
    assert sum(p.grad is None for p in named.values()) == int(g[f'train.B{B}.n_unused'])


def test_finetune_separate_test(ft_model):
    """forward_test(separate_test=True) (:128-148), one clip and two clips per sample, plus the retrieval metrics
    of the embeddings (accuracy.py:430-462) through clover_amd.evaluation."""
    from clover_amd.evaluation import recall_for_video_text_retrieval
    g = gutil.load('g_finetune.npz')
    batch = to_dev(cf.cf_batch(4, tag='ft_test'))
    with torch.no_grad():
        v, t = ft_model(batch['imgs'], None, return_loss=False, **{k: batch[k] for k in FT_AUX})
        imgs2 = batch['imgs'].reshape((2, 2) + tuple(batch['imgs'].shape[2:]))
        v2, t2 = ft_model.forward_test(imgs2, **{k: batch[k][:2] for k in FT_AUX})
    assert rel(v, g['test.clips1.visual_emb']) < 2e-2 and rel(t, g['test.clips1.text_emb']) < 2e-2
    assert rel(v2, g['test.clips2.visual_emb']) < 2e-2 and rel(t2, g['test.clips2.text_emb']) < 2e-2
    m = recall_for_video_text_retrieval(v, t)
    mr = recall_for_video_text_retrieval(g['test.clips1.visual_emb'], g['test.clips1.text_emb'])
    assert m == mr


def test_finetune_other_tasks_refuse():
    import clover_amd
    for task in ('video_qa', 'FIB', None):
        cfg = cf.tiny_finetune_cfg()
        cfg['task'] = task
        with pytest.raises(NotImplementedError):
            clover_amd.build_model(cfg)


# ----------------------------------------------------------------------------- BASELINE configs 2 and 4 at full size
FULL_GRAD_TOL = 8e-2          # max|grad - oracle| / max|oracle|; measured 0.8-4.7 % (round 4, incl. Swin-B 32 frames)
FULL_GRAD_KEYS = {
    'T': ['backbone.patch_embed.proj.weight', 'backbone.layers.2.blocks.3.attn.relative_position_bias_table',
          'backbone.layers.1.downsample.reduction.weight', 'backbone.layers.3.blocks.1.mlp.fc2.weight',
          'text_backbone.bert.encoder.layer.6.attention.self.query.weight',
          'multimodal_backbone.bert_encoder.layer.2.output.dense.weight', 'mlm_head.predictions.decoder.weight'],
    # Swin-B: the fc_in projection (1024 -> 768, cross_transformer.py:69-70) exists only here; stage 2 is 18 blocks deep
    'B': ['backbone.patch_embed.proj.weight', 'backbone.layers.2.blocks.11.attn.relative_position_bias_table',
          'backbone.layers.2.blocks.14.mlp.fc1.weight', 'backbone.layers.2.blocks.17.attn.qkv.weight',
          'backbone.layers.1.downsample.reduction.weight', 'backbone.layers.3.blocks.1.mlp.fc2.weight',
          'multimodal_backbone.fc_in.weight', 'multimodal_backbone.bert_encoder.layer.2.output.dense.weight',
          'text_backbone.bert.encoder.layer.6.attention.self.query.weight', 'mlm_head.predictions.decoder.weight'],
}


@pytest.mark.usefixtures('strict_own_gemm')
@pytest.mark.parametrize('variant,frames', [('T', 8), ('B', 16), ('B', 32)])
def test_full_size_step_matches_oracle(variant, frames):
    """BASELINE config 2 (VideoSwin-T, 8 frames), config 4 (VideoSwin-B: embed_dim 128, heads [4,8,16,32], the
    depth-18 stage, fc_in 1024 -> 768; 16 frames -> 392-token windows) and config 5's model + clip length (VideoSwin-B,
    32 frames: the (4,3,3)-shifted (8,7,7) windows only 32 frames reach, swin_transformer_3d.py:302-315; 816-token fusion
    sequences) + BERT-base + 3-layer fusion at the benchmark's shapes (224^2, 32 tokens, B = 2, seeded random init, eval
    mode so no dropout / DropPath): the five losses of the HIP step against the fp32 oracle on the host cores, and
    gradients that cross every encoder."""
    import clover_amd
    cfg, sd, batch, lv_ref, gref = gutil.full_size_oracle(variant, frames)
    m = clover_amd.build_model(cfg).eval()
    m.load_state_dict(sd)
    m = m.to(DEV)
    out = m.train_step({k: v.to(DEV) for k, v in batch.items()}, None)
    lv = out['log_vars']
    errs = {k: abs(lv[k] - lv_ref[k]) for k in LOSS_KEYS}
    print(f'full-size Swin-{variant} {frames}f loss errors', errs, {k: lv_ref[k] for k in LOSS_KEYS})
    for k in LOSS_KEYS:
        assert errs[k] <= LOSS_TOL[k], (k, lv[k], lv_ref[k])
    out['loss'].backward()
    named = dict(m.named_parameters())
    worst = {k: rel(named[k].grad, gref[k].numpy()) for k in FULL_GRAD_KEYS[variant]}
    print(f'full-size Swin-{variant} {frames}f grad rel errors', worst)
    for k, e in worst.items():
        assert e < FULL_GRAD_TOL, (k, e)


@pytest.mark.usefixtures('strict_own_gemm')
@pytest.mark.parametrize('mode', ['graph', 'eager'])
def test_bench_shapes_step_matches_reference(mode):
    """BASELINE config 2 at the BENCHMARK's batch (VideoSwin-T + BERT-base + 3-layer fusion, 8 clips x 8 frames x 224^2,
    32 tokens: the GEMM tile classes, split-K plans and weight-gradient groupings bench.py times — M = 200 704 / 50 176 /
    12 544 / 3 136 / 3 648 / 512 rows — differ from the B = 2 ones) THROUGH THE ENGINE (slab sinks, first-touch stores,
    deferred grouped weight gradients, phantom-padded MLM decoder; 'graph': the replayed hipGraphs the bench times),
    eval mode: six losses and the seven FULL_GRAD_KEYS['T'] gradients against the REAL reference's numbers for the same
    weights and batch (tests/golden/g_full_b8.npz, written by `make_goldens.py full8`; the oracle is pinned to the same
    file by tests/test_oracle_golden.py).  Every GEMM on the HIP kernels (strict_own_gemm)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import clover_amd
    from clover_amd.engine import CloverEngine
    g = gutil.load('g_full_b8.npz')
    torch.manual_seed(4321)
    m = clover_amd.build_model(bench.model_cfg('T', 8)).eval().to(DEV)
    batch = {k: v.to(DEV) for k, v in bench.synthetic_batch(8, 8, 32, seed=77).items()}
    eng = CloverEngine(m, batch, lr=1e-4, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 6)
    assert eng.first_touch_params > 100e6                       # the first-touch stores are part of what is checked
    if mode == 'graph':
        eng.capture(batch)
        out = eng._graphed_forward_backward(batch)
    else:
        eng._ft.done.clear()
        out = m.train_step(batch, None)
        eng._backward(lambda: out['loss'].backward())
    eng.finish_backward()
    torch.cuda.synchronize()
    lv = {k: float(out['log_vars'][k]) for k in LOSS_KEYS}
    errs = {k: abs(lv[k] - float(g[f'T8.{k}'])) for k in LOSS_KEYS}
    print(f'bench-shape (B = 8, {mode}) loss errors', errs)
    for k in LOSS_KEYS:
        assert errs[k] <= LOSS_TOL[k], (k, lv[k], float(g[f'T8.{k}']))
    named = dict(m.named_parameters())
    # (the slabs hold the gradients times the loss scale of the 16-bit backward: 1024 in the fp16 build, 1 in the bf16 one)
    worst = {k: rel_packed(g, f'T8.grad.{k}', named[k].grad / eng.loss_scale) for k in FULL_GRAD_KEYS['T']}
    print(f'bench-shape (B = 8, {mode}) grad rel errors', worst)
    for k, e in worst.items():
        assert e < FULL_GRAD_TOL, (k, e)
    from clover_amd import ops
    assert ops.MLM_DECODER_STATS['in_place'] > 0               # the padded d-scores buffer is contracted over in place


@pytest.mark.usefixtures('strict_own_gemm')
def test_train_mode_step_is_finite_and_learns():
    """BASELINE config 2 shapes with model.train(): hidden dropout 0.1, attention-probability dropout 0.1 and DropPath
    0.1 all active inside the HIP kernels.  Twenty optimizer steps on one fixed batch: every loss finite, and the total
    falls."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import clover_amd
    from clover_amd.engine import CloverEngine
    torch.manual_seed(11)
    m = clover_amd.build_model(bench.model_cfg('T', 8)).to(DEV)
    m.train()
    batch = {k: v.to(DEV) for k, v in bench.synthetic_batch(4, 8, 32, seed=5).items()}
    eng = CloverEngine(m, batch, lr=1e-4, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 6)
    hist = []
    for _ in range(20):
        lv = eng.step(batch)['log_vars']
        vals = {k: float(v) for k, v in lv.items()}
        assert all(np.isfinite(v) for v in vals.values()), vals
        hist.append(vals['loss'])
    print('train-mode loss trajectory', [round(h, 3) for h in hist])
    assert min(hist[-5:]) < hist[0] - 1.0, hist              # dropout-noisy, 4 samples: the tail dips below the start
    m.eval()


@pytest.mark.usefixtures('strict_own_gemm')
@pytest.mark.parametrize('frames', [16, 32])
def test_long_clip_losses_match_oracle(frames):
    """16- and 32-frame clips (BASELINE configs 4 / 5 clip lengths; windows of 392 tokens; fusion sequences of 424 tokens
    on the fused kernels and of 816 tokens on the unfused GEMM + row-softmax path): the five losses of the HIP step
    against the fp32 oracle, Swin-T + BERT-base, B = 2, eval mode."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import clover_amd
    from oracle import model as om
    torch.manual_seed(97 + frames)
    cfg = bench.model_cfg('T', frames)
    m = clover_amd.build_model(cfg).eval()
    P = {k: v.detach().float() for k, v in m.state_dict().items() if 'relative_position_index' not in k}
    batch = bench.synthetic_batch(2, frames, 32, seed=78)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        _, lv_ref = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
    m = m.to(DEV)
    with torch.no_grad():
        lv = m.train_step({k: v.to(DEV) for k, v in batch.items()}, None)['log_vars']
    errs = {k: abs(lv[k] - lv_ref[k]) for k in LOSS_KEYS}
    print(f'{frames}-frame loss errors', errs)
    for k in LOSS_KEYS:
        assert errs[k] <= LOSS_TOL[k], (k, lv[k], lv_ref[k])


@pytest.mark.parametrize('size', [96, 80, 104])
def test_swin_padded_sizes(model, size):
    """Inputs whose token grids are not multiples of the window (24, 20, 26 tokens a side: window padding,
    swin_transformer_3d.py:452-457,478-479) and, for 104, odd at the merge (13 tokens: PatchMerging's zero pad
    :531-533): forward and two gradients against the oracle."""
    from oracle import model as om
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in cf.cf_state(gutil.manifest()).items()}
    ocfg = cf.oracle_cfg_from(cf.tiny_model_cfg())
    x = cf.cf_float(f'pad.{size}', (2, 3, 4, size, size), 1.7)
    ref = om.swin_forward(P, 'backbone.', x, ocfg['backbone'])
    w = cf.cf_float(f'pad.w.{size}', tuple(ref.shape), 1.0)
    (ref * w).sum().backward()
    model.zero_grad(set_to_none=True)
    y = model.backbone(x.to(DEV))
    assert y.shape == ref.shape
    assert rel(y, ref.detach().numpy()) < 3e-2, rel(y, ref.detach().numpy())
    (y.float() * w.to(DEV)).sum().backward()
    named = dict(model.named_parameters())
    for k in ['backbone.layers.0.blocks.1.attn.relative_position_bias_table', 'backbone.layers.0.downsample.reduction.weight',
              'backbone.patch_embed.proj.weight']:
        e = rel(named[k].grad, P[k].grad.numpy())
        assert e < 4e-2, (k, e)
    model.zero_grad(set_to_none=True)


def test_use_checkpoint_recomputes_and_matches():
    """``use_checkpoint=True`` (swin_transformer_3d.py:494-503 wraps both halves of a block in checkpoint.checkpoint):
    the blocks' activations are recomputed in the backward.  Same losses and the same gradients as without it (the
    recompute is the same kernels on the same inputs: bit-equal), through plain autograd and through the engine's sinks."""
    import clover_amd
    grads, losses = {}, {}
    for ck in (False, True):
        cfg = cf.tiny_model_cfg()
        cfg['backbone']['use_checkpoint'] = ck
        m = clover_amd.build_model(cfg)
        m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False)
        m = m.to(DEV).eval()
        assert all(layer.use_checkpoint == ck for layer in m.backbone.layers)
        out = m.train_step(to_dev(cf.cf_batch(2, tag='ckpt')), None)
        out['loss'].backward()
        losses[ck] = {k: float(v) for k, v in out['log_vars'].items()}
        grads[ck] = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    for k in losses[False]:           # the focal / InfoNCE sums are atomically accumulated: equal up to fp32 summation order
        assert abs(losses[True][k] - losses[False][k]) <= 1e-5 * max(1.0, abs(losses[False][k])), (k, losses[True], losses[False])
    assert grads[True].keys() == grads[False].keys()
    for n in grads[False]:
        a, b = grads[True][n].float(), grads[False][n].float()
        # atomically accumulated gradients (LayerNorm / table partial sums) differ by summation order only
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-9, n
