"""Pins the oracle (oracle/) against goldens produced by the reference itself
(tests/golden/make_goldens.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

import closed_form as cf
import gutil
from oracle import indexing as ix
from oracle import model as om


# ------------------------------------------------------------------ G-idx (bit-exact)
def test_relative_position_index():
    g = gutil.load('g_idx.npz')
    for ws in [(8, 7, 7), (4, 7, 7), (2, 7, 7)]:
        assert np.array_equal(ix.relative_position_index(ws), g['rpi_%d_%d_%d' % ws])
    # the [:N,:N] slice of the configured (8,7,7) index is NOT the native (4,7,7) index: +676
    full = ix.relative_position_index((8, 7, 7))
    assert np.array_equal(full[:196, :196], g['rpi_4_7_7'] + 676)


FS = [(4, 56, 56), (8, 56, 56), (16, 56, 56), (2, 28, 28), (2, 14, 14), (4, 7, 7), (16, 7, 7), (16, 14, 14), (3, 10, 12)]


@pytest.mark.parametrize('fs', FS)
def test_window_geometry(fs):
    g = gutil.load('g_idx.npz')
    tag = '%d_%d_%d' % fs
    ws, ss = ix.get_window_size(fs, (8, 7, 7), (4, 3, 3))
    assert list(ws) + list(ss) == g['gws_' + tag].tolist()
    Dp, Hp, Wp = [int(np.ceil(f / w)) * w for f, w in zip(fs, ws)]
    m = ix.compute_mask(Dp, Hp, Wp, ws, ss)
    assert set(np.unique(m).tolist()) <= {0.0, -100.0}
    assert np.array_equal((m != 0).astype(np.int8), g['mask_' + tag])
    rid = ix.window_region_ids(Dp, Hp, Wp, ws, ss)
    assert np.array_equal((rid[:, :, None] != rid[:, None, :]).astype(np.int8), g['mask_' + tag])
    assert np.array_equal(ix.shifted_window_token_index(Dp, Hp, Wp, ws, ss), g['part_' + tag])


def test_blend_merge_mlm_indices():
    g = gutil.load('g_idx.npz')
    w = ix.mask_blend_weight(g['blend_mask'], 2, 28, 28)
    assert np.array_equal(w.astype(np.int8), g['blend_w'])
    assert np.array_equal(ix.patch_merging_gather(g['merge_in']), g['merge_cat'])
    tid, lab = g['ssl_token_ids'][:, 0], g['ssl_mlm_label'][:, 0]
    assert np.array_equal(ix.input_ssl_ids(tid, lab), g['ssl_ids'])
    rows, labels = ix.mlm_rows(lab)
    assert np.array_equal(rows, g['mlm_rows']) and np.array_equal(labels, g['mlm_row_labels'])


# ------------------------------------------------------------------ float sets
@pytest.fixture(scope='module')
def P():
    return cf.cf_state(gutil.manifest())


@pytest.fixture(scope='module')
def ocfg():
    return cf.oracle_cfg_from(cf.tiny_model_cfg())


def test_swin(P, ocfg):
    g = gutil.load('g_swin.npz')
    Pg = {k: v.clone().requires_grad_() for k, v in P.items() if k.startswith('backbone.')}
    batch = cf.cf_batch(2, tag='swin')
    x, vm = batch['imgs'][:, 0], batch['v_token_mask']
    taps = {}
    y = om.swin_forward(Pg, 'backbone.', x, ocfg['backbone'], taps=taps)
    for k, v in taps.items():
        gutil.assert_packed(g, 'clean.' + k, v if k != 'patch_embed' else v)
    gutil.assert_close(y, g['clean.out'], name='clean.out')
    taps = {}
    ym, w = om.swin_forward(Pg, 'backbone.', x.clone(), ocfg['backbone'], vm, taps=taps)
    for k, v in taps.items():
        gutil.assert_packed(g, 'masked.' + k, v)
    gutil.assert_close(ym, g['masked.out'], name='masked.out')
    assert np.array_equal(w.numpy().astype(np.int8), g['masked.w'])
    gw = cf.cf_float('swin.gw', tuple(y.shape), 1.0)
    ((y * gw).sum() + (ym * gw.flip(0)).sum()).backward()
    for k in [n[5:-4] for n in g.files if n.startswith('grad.') and n.endswith('.sub')]:
        gutil.assert_packed(g, 'grad.' + k, Pg['backbone.' + k].grad, rtol=2e-4)


def test_bert_and_fusion(P, ocfg):
    g = gutil.load('g_bert_fuse.npz')
    b = cf.cf_batch(3, tag='bf')
    ids, mask = b['token_ids'][:, 0], b['input_mask'][:, 0]
    t = om.bert_forward(P, 'text_backbone.', ids, mask, ocfg['bert'])
    gutil.assert_close(t, g['bert.last_hidden_state'], name='bert')
    vt = cf.cf_float('bf.vt', (3, 2, 196, 96), 1.0)
    f = om.fusion_forward(P, 'multimodal_backbone.', vt, mask, t, ocfg['fusion'])
    gutil.assert_close(f['t_last_hidden_state'], g['fuse.t_last_hidden_state'], name='fuse.t')
    gutil.assert_packed(g, 'fuse.v_last_hidden_state', f['v_last_hidden_state'])


def test_heads_and_losses(P):
    g = gutil.load('g_heads_loss.npz')
    vis = cf.cf_float('hl.vis', (4, 96, 2, 14, 14), 1.0)
    txt = cf.cf_float('hl.txt', (4, 16, 128), 1.0)
    gutil.assert_close(om.nce_mm_forward_vision(P, 'ssl_head.', vis), g['mm.vision'])
    gutil.assert_close(om.nce_mm_forward_vision(P, 'ssl_head.', vis[:1]), g['mm.vision_b1'])
    gutil.assert_close(om.nce_mm_forward_text(P, 'ssl_head.', txt), g['mm.text'])
    row = cf.cf_float('hl.row', (4, 128), 1.0)
    gutil.assert_close(om.nce_vision_head(P, 'mlm_ssl_V_head.', row), g['V.head'])
    gutil.assert_close(om.nce_text_head(P, 'mlm_ssl_T_head.', row), g['T.head'])
    gutil.assert_close(om.mlm_head(P, 'mlm_head.', txt[:2]), g['mlm.scores'])
    logits = cf.cf_float('hl.logits', (7, 1024), 4.0)
    tgt = cf.cf_int('hl.tgt', (7,), 0, 1024)
    gutil.assert_close(om.focal_loss_multiclass(logits, tgt), g['focal'])
    for G in [1, 2, 4, 8]:
        e = [cf.cf_float(f'hl.e{k}.{G}', (G, 128), 1.0) for k in range(4)]
        l = om.exclusive_nce_rank_loss(*e, gather=False)
        gutil.assert_close(l['nce_loss'], g[f'nce.G{G}.nce_loss'], atol=2e-5, name=f'nce G{G}')
        gutil.assert_close(l['rank_t_tm_loss'], g[f'nce.G{G}.rank_t_tm_loss'], name=f'rank G{G}')
    e = [cf.cf_float(f'hl.e{k}.4', (4, 128), 1.0).requires_grad_() for k in range(4)]
    l = om.exclusive_nce_rank_loss(*e, gather=False)
    (l['nce_loss'] + l['rank_t_tm_loss']).backward()
    for k in range(4):
        gutil.assert_close(e[k].grad, g[f'nce.G4.grad{k}'], rtol=2e-4, name=f'grad{k}')


@pytest.mark.parametrize('B', [1, 2, 4])
def test_step_config1(P, ocfg, B):
    """BASELINE config 1: full forward_train -> 5 losses + total, plus selected grads."""
    g = gutil.load('g_step.npz')
    Pg = {k: v.clone().requires_grad_() for k, v in P.items()}
    batch = cf.cf_batch(B, tag=f'step{B}')
    losses = om.forward_train(Pg, batch, ocfg, gather=False)
    loss, lv = om.parse_losses(losses)
    for k in ['mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss']:
        assert abs(lv[k] - float(g[f'B{B}.{k}'])) <= 2e-4 * max(1.0, abs(float(g[f'B{B}.{k}']))), (k, lv[k], float(g[f'B{B}.{k}']))
    loss.backward()
    for k in [n[len(f'B{B}.grad.'):-4] for n in g.files if n.startswith(f'B{B}.grad.') and n.endswith('.sub')]:
        gutil.assert_packed(g, f'B{B}.grad.{k}', Pg[k].grad, rtol=5e-3, atol=1e-6)  # fp32 summation-order noise through the /0.05 logits
    unused = sorted(k for k, p in Pg.items() if p.grad is None)
    assert unused == gutil.unused_params()


# ------------------------------------------------------------------ retrieval fine-tuning
def _recall_inputs(N, D):
    ve = cf.cf_float(f'recall.N{N}.v', (N, D), 1.0).numpy()
    te = (0.35 * ve + cf.cf_float(f'recall.N{N}.t', (N, D), 1.0).numpy()).astype(np.float32)
    if N == 7:
        te[3] = 0
    return ve, te


def test_norm_softmax_loss_and_recall():
    g = gutil.load('g_finetune.npz')
    for cos in (False, True):
        for G in (1, 2, 4, 8, 33):
            for Dm in (128, 50):
                tag = f'loss.cos{int(cos)}.G{G}.D{Dm}'
                v = cf.cf_float(tag + '.v', (G, Dm), 1.3).requires_grad_()
                t = cf.cf_float(tag + '.t', (G, Dm), 0.9, 0.1).requires_grad_()
                loss = om.norm_softmax_loss(v, t, temperature=0.05 if cos else 0.07, cos_sim=cos, gather=False)
                loss.backward()
                gutil.assert_close(loss, g[tag], atol=2e-5, name=tag)
                gutil.assert_close(v.grad, g[tag + '.dv'], rtol=2e-4, atol=1e-6, name=tag + '.dv')
                gutil.assert_close(t.grad, g[tag + '.dt'], rtol=2e-4, atol=1e-6, name=tag + '.dt')
        v = cf.cf_float('loss.zero.v', (4, 32), 1.0)
        v[2] = 0
        t = cf.cf_float('loss.zero.t', (4, 32), 1.0).requires_grad_()
        loss = om.norm_softmax_loss(v, t, temperature=0.07, cos_sim=cos, gather=False)
        loss.backward()
        gutil.assert_close(loss, g[f'loss.zero.cos{int(cos)}'], name='zero row')
        gutil.assert_close(t.grad, g[f'loss.zero.cos{int(cos)}.dt'], rtol=2e-4, atol=1e-6, name='zero row dt')
    x = cf.cf_float('loss.sim', (6, 6), 9.0).requires_grad_()
    loss = om.norm_softmax_loss(sim_mat=x)
    loss.backward()
    gutil.assert_close(loss, g['loss.sim'])
    gutil.assert_close(x.grad, g['loss.sim.dx'], rtol=2e-4, atol=1e-6)
    for N, D in ((1, 8), (7, 16), (50, 32), (200, 64)):
        np.testing.assert_allclose(om.recall_metrics(*_recall_inputs(N, D)), g[f'recall.N{N}'], rtol=0, atol=1e-9)


@pytest.mark.parametrize('B', [2, 4])
def test_finetune_retrieval_step(P, ocfg, B):
    g = gutil.load('g_finetune.npz')
    Pg = {k: v.clone().requires_grad_() for k, v in P.items()}
    losses = om.finetune_forward_train(Pg, cf.cf_batch(B, tag=f'ft{B}'), ocfg, gather=False)
    loss, lv = om.parse_losses(losses)
    for k in ['retrieval_nce_loss', 'loss']:
        ref = float(g[f'train.B{B}.{k}'])
        assert abs(lv[k] - ref) <= 2e-4 * max(1.0, abs(ref)), (k, lv[k], ref)
    loss.backward()
    for k in [n[len(f'train.B{B}.grad.'):-4] for n in g.files if n.startswith(f'train.B{B}.grad.') and n.endswith('.sub')]:
        gutil.assert_packed(g, f'train.B{B}.grad.{k}', Pg[k].grad, rtol=5e-3, atol=1e-6)


def test_finetune_separate_test(P, ocfg):
    g = gutil.load('g_finetune.npz')
    batch = cf.cf_batch(4, tag='ft_test')
    with torch.no_grad():
        v, t = om.finetune_embeddings(P, batch, ocfg)
        gutil.assert_close(v, g['test.clips1.visual_emb'], rtol=2e-4, name='visual_emb')
        gutil.assert_close(t, g['test.clips1.text_emb'], rtol=2e-4, name='text_emb')
        b2 = dict(batch)
        b2['imgs'] = batch['imgs'].reshape((2, 2) + tuple(batch['imgs'].shape[2:]))
        for k in ('token_ids', 'input_mask'):
            b2[k] = batch[k][:2]
        v, t = om.finetune_embeddings(P, b2, ocfg)
        gutil.assert_close(v, g['test.clips2.visual_emb'], rtol=2e-4, name='visual_emb clips2')
        gutil.assert_close(t, g['test.clips2.text_emb'], rtol=2e-4, name='text_emb clips2')


# ------------------------------------------------------------------ G-full: the REAL reference at benchmark shapes
def _packed_rel(g, name, t):
    sub, stats = gutil.packed(t)
    gsub = g[name + '.sub'].astype(np.float64)
    assert g[name + '.stats'][2] == stats[2]
    return np.abs(sub - gsub).max() / max(np.abs(gsub).max(), 1e-20)


def test_oracle_matches_reference_at_config2_shapes():
    """g_full.npz (make_goldens.py full): the real reference run at BASELINE config 2's shapes (VideoSwin-T + BERT-base, 8 f x
    224^2, 32 tokens, B = 2) on the seed-4321 weights and the bench's synthetic batch (seed 77) — exactly what
    tests/gutil.py::full_size_oracle hands the GPU tests.  The oracle must reproduce its six losses (2e-5 absolute on
    values of 5-30: fp32 summation order) and the gradients the full-size GPU tests read (1e-3 of max)."""
    g = gutil.load('g_full.npz')
    cfg, sd, batch, lv, grads = gutil.full_size_oracle('T', 8)
    for k in ('mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss'):
        assert abs(lv[k] - float(g[f'T8.{k}'])) <= 2e-5 * max(1.0, abs(float(g[f'T8.{k}']))), (k, lv[k], float(g[f'T8.{k}']))
    names = [n[len('T8.grad.'):-4] for n in g.files if n.startswith('T8.grad.') and n.endswith('.sub')]
    assert len(names) >= 5
    for n in names:
        e = _packed_rel(g, f'T8.grad.{n}', grads[n])
        assert e < 1e-3, (n, e)


def test_oracle_matches_reference_at_config5_model_and_clip_length():
    """The same pin for VideoSwin-B + BERT-base at 32 frames (config 5's model and clip length; the (4,3,3)-shifted
    (8,7,7) windows, 816-token fusion sequences): the oracle's forward (no backward here: the CPU suite stays short — the
    GPU tests take the gradients from the oracle, whose backward is torch autograd over the same graph) against the
    reference's six losses."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import clover_amd
    g = gutil.load('g_full.npz')
    torch.manual_seed(4321)
    cfg = bench.model_cfg('B', 32)
    m = clover_amd.build_model(cfg).eval()
    P = {k: v.detach().float() for k, v in m.state_dict().items() if 'relative_position_index' not in k}
    del m
    batch = bench.synthetic_batch(2, 32, 32, seed=77)
    with torch.no_grad():
        _, lv = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
    for k in ('mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss'):
        assert abs(float(lv[k]) - float(g[f'B32.{k}'])) <= 2e-5 * max(1.0, abs(float(g[f'B32.{k}']))), (k, float(lv[k]), float(g[f'B32.{k}']))


def test_oracle_matches_reference_at_the_benchmark_batch():
    """g_full_b8.npz (make_goldens.py full8): the real reference at BASELINE config 2 with the BENCHMARK's batch of 8 clips
    (the shapes bench.py times; tests/test_step_gpu.py::test_bench_shapes_step_matches_reference compares the HIP step with
    this file directly).  The oracle's forward reproduces the six losses — the bench's `loss_abs_err_vs_oracle` and the
    cpu_baseline leg run the oracle at these widths."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import clover_amd
    g = gutil.load('g_full_b8.npz')
    torch.manual_seed(4321)
    cfg = bench.model_cfg('T', 8)
    m = clover_amd.build_model(cfg).eval()
    P = {k: v.detach().float() for k, v in m.state_dict().items() if 'relative_position_index' not in k}
    del m
    batch = bench.synthetic_batch(8, 8, 32, seed=77)
    with torch.no_grad():
        _, lv = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
    for k in ('mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss'):
        assert abs(float(lv[k]) - float(g[f'T8.{k}'])) <= 2e-5 * max(1.0, abs(float(g[f'T8.{k}']))), (k, float(lv[k]), float(g[f'T8.{k}']))


@pytest.mark.parametrize('B', [2, 4])
def test_oracle_step_at_mid_widths(B):
    """g_mid.npz: the reference at VideoSwin-T's stage widths + BERT-tiny (cf.mid_model_cfg) — the oracle reproduces its
    losses and the 23 stored gradients (the GPU tests compare the HIP kernels with the same file)."""
    g = gutil.load('g_mid.npz')
    cfg = cf.mid_model_cfg()
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in cf.cf_state(gutil.manifest('mid')).items()}
    losses = om.forward_train(P, cf.cf_batch(B, tag=f'mid.step{B}'), cf.oracle_cfg_from(cfg), gather=False)
    loss, lv = om.parse_losses(losses)
    for k, v in lv.items():
        assert abs(float(v) - float(g[f'B{B}.{k}'])) <= 2e-4 * max(1.0, abs(float(g[f'B{B}.{k}']))), (k, float(v))
    loss.backward()
    for n in [n[len(f'B{B}.grad.'):-4] for n in g.files if n.startswith(f'B{B}.grad.') and n.endswith('.sub')]:
        gutil.assert_packed(g, f"B{B}.grad.{n}", P[n].grad, rtol=5e-3, atol=1e-6)


# ------------------------------------------------------------------ loss scaler (exact)
def test_loss_scaler_matches_reference_trajectories():
    """oracle/loss_scaler.py against the reference's own LossScaler (fp16_utils.py:285-389) driven by the same overflow
    flags: the scale before every iteration and the final state_dict, exactly (powers of two)."""
    from oracle.loss_scaler import LossScaler
    g = gutil.load('g_scaler.npz')
    for name, kw, flags in cf.scaler_cases():
        assert g[name + '.flags'].tolist() == list(flags)
        sc = LossScaler(**kw)
        for i, f in enumerate(flags):
            assert float(sc.loss_scale) == float(g[name + '.scales'][i]), (name, i)
            sc.update_scale(bool(f))
        st = sc.state_dict()
        assert [float(st[k]) for k in ('cur_scale', 'cur_iter', 'last_overflow_iter', 'scale_factor', 'scale_window')] \
            == g[name + '.final'].tolist(), name
