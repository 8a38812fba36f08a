"""Golden-vector generator — BUILD-CONTAINER ONLY.

Runs the REAL reference (imported from /root/reference through ref_harness.py) on
closed-form weights/inputs (closed_form.py) and writes small .npz fixtures next to
this file.  The parity tests (and the GPU box) only ever read the fixtures.

    python tests/golden/make_goldens.py            # all sets
    python tests/golden/make_goldens.py dist       # one set

Sets: scaler (LossScaler trajectories), idx, swin, bert_fuse, heads_loss, step, mid (VideoSwin-T stage widths: the HIP GEMMs' shapes), dist, inflate, test, finetune, full (benchmark shapes:
~10 minutes and ~40 GB on 8 cores; not part of the default list), full8 (config 2 at the benchmark's batch of 8).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import closed_form as cf  # noqa: E402
import ref_harness as H  # noqa: E402

SCRATCH = '/tmp/clover_golden_scratch'
MAXSUB = 4096


def pack(out, name, t):
    """Store a strided subsample (<= MAXSUB values) + [sum, l2, numel] of a tensor."""
    a = t.detach().cpu().double().numpy().reshape(-1)
    stride = max(1, a.size // MAXSUB)
    out[name + '.sub'] = a[::stride][:MAXSUB].astype(np.float32)
    out[name + '.stats'] = np.array([a.sum(), np.sqrt((a * a).sum()), a.size], dtype=np.float64)


def full(out, name, t):
    out[name] = t.detach().cpu().numpy()


def save(fname, out):
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **out)
    print(f'wrote {fname}: {len(out)} arrays, {os.path.getsize(path) / 1024:.1f} KiB')


def ref_model(drop=0.0, cfg=None):
    H.make_bert_dir(SCRATCH, hidden=cf.TINY_BERT['hidden_size'], layers=cf.TINY_BERT['num_hidden_layers'],
                    heads=cf.TINY_BERT['num_attention_heads'], inter=cf.TINY_BERT['intermediate_size'],
                    vocab=cf.TINY_BERT['vocab_size'], max_pos=cf.TINY_BERT['max_position_embeddings'])
    cfg = cfg if cfg is not None else cf.tiny_model_cfg(drop)
    # the reference classes take bert_config only through **kwargs (ignored); MLMHead has no kwargs
    m = H.build_reference_model(cfg, SCRATCH)
    manifest = {k: list(v.shape) for k, v in m.state_dict().items()}
    sd = cf.cf_state(manifest)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert all('relative_position_index' in k for k in missing), missing
    assert not unexpected, unexpected
    return m, manifest


# --------------------------------------------------------------------------- #
def gen_idx():
    H.install_shims()
    import mmaction.models.backbones.swin_transformer_3d as S
    out = {}
    for ws in [(8, 7, 7), (4, 7, 7), (2, 7, 7)]:
        att = S.WindowAttention3D(6, ws, 3)
        out['rpi_%d_%d_%d' % ws] = att.relative_position_index.numpy()
    cfg_ws, cfg_ss = (8, 7, 7), (4, 3, 3)
    for fs in [(4, 56, 56), (8, 56, 56), (16, 56, 56), (2, 28, 28), (2, 14, 14), (4, 7, 7), (16, 7, 7), (16, 14, 14), (3, 10, 12)]:
        ws, ss = S.get_window_size(fs, cfg_ws, cfg_ss)
        tag = '%d_%d_%d' % fs
        out['gws_' + tag] = np.array(list(ws) + list(ss), dtype=np.int64)
        Dp, Hp, Wp = [int(np.ceil(f / w)) * w for f, w in zip(fs, ws)]
        S.compute_mask.cache_clear()
        m = S.compute_mask(Dp, Hp, Wp, ws, ss, torch.device('cpu'), torch.float32)
        out['mask_' + tag] = (m != 0).numpy().astype(np.int8)
        assert set(np.unique(m.numpy()).tolist()) <= {0.0, -100.0}
        # roll + partition and reverse + roll applied to arange
        ar = torch.arange(Dp * Hp * Wp, dtype=torch.int64).view(1, Dp, Hp, Wp, 1)
        sh = torch.roll(ar, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3)) if any(ss) else ar
        part = S.window_partition(sh, ws)
        out['part_' + tag] = part[..., 0].numpy()
        rev = S.window_reverse(part.view(-1, *(ws + (1,))), ws, 1, Dp, Hp, Wp)
        rev = torch.roll(rev, shifts=ss, dims=(1, 2, 3)) if any(ss) else rev
        assert torch.equal(rev, ar)
    # mask-blend weight via a real forward with constant weights
    vm = cf.cf_batch(3)['v_token_mask'][:, 0][:, None]
    sw = S.SwinTransformer3D(embed_dim=48, depths=[2], num_heads=[3], mask_token=True, pretrained2d=False)
    _, w = sw(torch.zeros(3, 3, 4, 112, 112), vm)
    out['blend_mask'] = vm.numpy()
    out['blend_w'] = w.numpy().astype(np.int8)
    # patch merging gather order on arange (norm/reduction bypassed by capturing the cat)
    pm = S.PatchMerging(dim=1)
    x = torch.arange(2 * 2 * 5 * 6, dtype=torch.float32).view(2, 2, 5, 6, 1)
    grabbed = {}
    pm.norm.register_forward_pre_hook(lambda mod, a: grabbed.__setitem__('x', a[0].clone()))
    pm(x)
    out['merge_in'] = x.numpy()
    out['merge_cat'] = grabbed['x'].numpy()
    b = cf.cf_batch(4)
    out['ssl_token_ids'] = b['token_ids'].numpy()
    out['ssl_mlm_label'] = b['mlm_label'].numpy()
    tid, lab = b['token_ids'][:, 0], b['mlm_label'][:, 0]
    out['ssl_ids'] = torch.where(lab == -100, tid.clone(), lab.clone()).numpy()
    idx = torch.where(lab.reshape(-1) != -100)
    out['mlm_rows'] = idx[0].numpy()
    out['mlm_row_labels'] = lab.reshape(-1)[idx].numpy()
    save('g_idx.npz', out)


def gen_swin():
    m, manifest = ref_model()
    m.eval()
    bb = m.backbone
    out = {}
    batch = cf.cf_batch(2, tag='swin')
    x = batch['imgs'][:, 0]
    vm = batch['v_token_mask']
    taps = {}
    hooks = [bb.patch_embed.register_forward_hook(lambda mod, a, o: taps.__setitem__('patch_embed', o))]
    for i, layer in enumerate(bb.layers):
        for j, blk in enumerate(layer.blocks):
            hooks.append(blk.register_forward_hook(
                lambda mod, a, o, k=f'layers.{i}.blocks.{j}': taps.__setitem__(k, o)))
    y = bb(x)
    for k, v in taps.items():
        pack(out, 'clean.' + k, v)
    full(out, 'clean.out', y)
    taps.clear()
    ym, w = bb(x.clone(), vm)
    for k, v in taps.items():
        pack(out, 'masked.' + k, v)
    full(out, 'masked.out', ym)
    out['masked.w'] = w.numpy().astype(np.int8)
    for h in hooks:
        h.remove()
    # grads of a closed-form scalar functional of both outputs
    gw = cf.cf_float('swin.gw', tuple(y.shape), 1.0)
    bb.zero_grad()
    ((y * gw).sum() + (ym * gw.flip(0)).sum()).backward()
    for k in ['layers.0.blocks.1.attn.qkv.weight', 'layers.0.blocks.1.attn.relative_position_bias_table',
              'layers.1.blocks.0.attn.relative_position_bias_table', 'mask_token', 'patch_embed.proj.weight',
              'layers.0.downsample.reduction.weight', 'layers.1.blocks.1.mlp.fc1.weight', 'norm.weight']:
        p = dict(bb.named_parameters())[k]
        pack(out, 'grad.' + k, p.grad)
    save('g_swin.npz', out)
    with open(os.path.join(HERE, 'manifest_tiny.json'), 'w') as f:
        json.dump(manifest, f, indent=0)


def gen_bert_fuse():
    m, _ = ref_model()
    m.eval()
    out = {}
    b = cf.cf_batch(3, tag='bf')
    ids, mask = b['token_ids'][:, 0], b['input_mask'][:, 0]
    t = m.text_backbone(ids, mask)['last_hidden_state']
    full(out, 'bert.last_hidden_state', t)
    vt = cf.cf_float('bf.vt', (3, 2, 196, 96), 1.0)
    f = m.multimodal_backbone(visual_token=vt, text_input_mask=mask, text_input_embeds=t)
    full(out, 'fuse.t_last_hidden_state', f['t_last_hidden_state'])
    pack(out, 'fuse.v_last_hidden_state', f['v_last_hidden_state'])
    save('g_bert_fuse.npz', out)


def gen_heads_loss():
    m, _ = ref_model()
    m.eval()
    H.init_dist_single()
    out = {}
    vis = cf.cf_float('hl.vis', (4, 96, 2, 14, 14), 1.0)
    txt = cf.cf_float('hl.txt', (4, 16, 128), 1.0)
    full(out, 'mm.vision', m.ssl_head.forward_vision(vis))
    full(out, 'mm.vision_b1', m.ssl_head.forward_vision(vis[:1]))
    full(out, 'mm.text', m.ssl_head.forward_text(txt))
    row = cf.cf_float('hl.row', (4, 128), 1.0)
    full(out, 'V.head', m.mlm_ssl_V_head(row))
    full(out, 'T.head', m.mlm_ssl_T_head(row))
    full(out, 'mlm.scores', m.mlm_head(txt[:2]))
    logits = cf.cf_float('hl.logits', (7, 1024), 4.0)
    tgt = cf.cf_int('hl.tgt', (7,), 0, 1024)
    full(out, 'focal', m.mlm_loss_func(logits, tgt))
    for G in [1, 2, 4, 8]:
        e = [cf.cf_float(f'hl.e{k}.{G}', (G, 128), 1.0) for k in range(4)]
        l = m.ssl_loss(*e)
        full(out, f'nce.G{G}.nce_loss', l['nce_loss'])
        full(out, f'nce.G{G}.rank_t_tm_loss', l['rank_t_tm_loss'])
    # gradient of the loss wrt the four embeddings at G=4
    e = [cf.cf_float(f'hl.e{k}.4', (4, 128), 1.0).requires_grad_() for k in range(4)]
    l = m.ssl_loss(*e)
    (l['nce_loss'] + l['rank_t_tm_loss']).backward()
    for k in range(4):
        full(out, f'nce.G4.grad{k}', e[k].grad)
    save('g_heads_loss.npz', out)


GRAD_KEYS = ['backbone.patch_embed.proj.weight', 'backbone.mask_token',
             'backbone.layers.0.blocks.1.attn.relative_position_bias_table',
             'backbone.layers.1.blocks.1.attn.qkv.weight',
             'text_backbone.bert.embeddings.word_embeddings.weight',
             'text_backbone.bert.encoder.layer.1.attention.self.query.weight',
             'multimodal_backbone.vis_space_pos', 'multimodal_backbone.fc_in.weight',
             'multimodal_backbone.bert_encoder.layer.0.intermediate.dense.weight',
             'mlm_head.predictions.decoder.weight', 'ssl_head.img_projector.0.weight',
             'mlm_ssl_V_head.img_fc1.weight', 'mlm_ssl_T_head.fc2.bias']



# --------------------------------------------------------------------------- #
def gen_inflate():
    """The reference's own SwinTransformer3D.inflate_weights (:130-181) on the synthetic 2-D checkpoint."""
    H.install_shims()
    import mmaction.models.backbones.swin_transformer_3d as R
    captured = {}
    R.load_state_dict = lambda module, sd, strict, logger: captured.update(sd)      # the (stubbed) mmcv loader
    path = os.path.join(SCRATCH, 'swin2d_synth.pth')
    os.makedirs(SCRATCH, exist_ok=True)
    torch.save({'state_dict': cf.inflate_checkpoint_2d()}, path)
    cfg = cf.tiny_model_cfg()['backbone']
    bb = R.SwinTransformer3D(**{k: v for k, v in cfg.items() if k != 'type'})

    class _Log:
        def info(self, *a, **k): pass
        def warning(self, *a, **k): pass
    bb.pretrained = path
    bb.inflate_weights(_Log())
    out = {k: v.detach().cpu().numpy() for k, v in captured.items()}
    out['__keys__'] = np.array(sorted(captured.keys()))
    save('g_inflate.npz', out)

def _step(m, batch):
    aux = {k: batch[k] for k in cf.AUX}
    losses = m(batch['imgs'], batch['label'], return_loss=True, **aux)
    loss, log_vars = m._parse_losses(losses)
    return loss, log_vars


def gen_step():
    m, _ = ref_model()
    m.eval()
    H.init_dist_single()
    out = {}
    for B in [1, 2, 4]:
        batch = cf.cf_batch(B, tag=f'step{B}')
        m.zero_grad()
        loss, lv = _step(m, batch)
        loss.backward()
        for k, v in lv.items():
            out[f'B{B}.{k}'] = np.float64(v)
        named = dict(m.named_parameters())
        for k in GRAD_KEYS:
            pack(out, f'B{B}.grad.{k}', named[k].grad)
        unused = [k for k, p in named.items() if p.grad is None]
        out[f'B{B}.n_unused'] = np.int64(len(unused))
        if B == 1:
            with open(os.path.join(HERE, 'unused_params_tiny.json'), 'w') as f:
                json.dump(sorted(unused), f, indent=0)
    # train_step contract
    batch = cf.cf_batch(2, tag='step2')
    o = m.train_step(batch, None)
    out['train_step.num_samples'] = np.int64(o['num_samples'])
    out['train_step.loss'] = np.float64(o['loss'].item())
    save('g_step.npz', out)


MID_GRAD_KEYS = ['backbone.patch_embed.proj.weight', 'backbone.mask_token',
                 'backbone.layers.0.blocks.0.attn.qkv.weight', 'backbone.layers.0.blocks.0.attn.proj.weight',
                 'backbone.layers.0.blocks.1.mlp.fc1.weight', 'backbone.layers.0.blocks.1.mlp.fc2.weight',
                 'backbone.layers.0.blocks.1.norm2.weight',
                 'backbone.layers.0.blocks.1.attn.relative_position_bias_table',
                 'backbone.layers.0.downsample.reduction.weight',
                 'backbone.layers.1.blocks.0.attn.qkv.weight', 'backbone.layers.1.blocks.1.attn.proj.weight',
                 'backbone.layers.1.blocks.0.mlp.fc1.weight', 'backbone.layers.1.blocks.1.mlp.fc2.weight',
                 'backbone.layers.1.blocks.1.mlp.fc2.bias',
                 'text_backbone.bert.encoder.layer.1.attention.self.query.weight',
                 'text_backbone.bert.encoder.layer.0.intermediate.dense.weight',
                 'text_backbone.bert.encoder.layer.1.output.dense.weight',
                 'multimodal_backbone.fc_in.weight',
                 'multimodal_backbone.bert_encoder.layer.0.intermediate.dense.weight',
                 'multimodal_backbone.bert_encoder.layer.1.attention.output.dense.weight',
                 'mlm_head.predictions.transform.dense.weight', 'mlm_head.predictions.decoder.weight',
                 'ssl_head.img_projector.0.weight']


def gen_mid():
    """cf.mid_model_cfg (VideoSwin-T's stage-0 / 1 widths + BERT-tiny: every Linear on the HIP GEMM kernels): the
    reference's backbone activations (per block, clean + masked), its step losses and 23 parameter gradients at B = 2, 4."""
    m, manifest = ref_model(cfg=cf.mid_model_cfg())
    m.eval()
    H.init_dist_single()
    out = {}
    with open(os.path.join(HERE, 'manifest_mid.json'), 'w') as f:
        json.dump(manifest, f, indent=0)
    bb = m.backbone
    batch = cf.cf_batch(2, tag='mid.swin')
    x, vm = batch['imgs'][:, 0], batch['v_token_mask']
    taps = {}
    hooks = [bb.patch_embed.register_forward_hook(lambda mod, a, o: taps.__setitem__('patch_embed', o))]
    for i, layer in enumerate(bb.layers):
        for j, blk in enumerate(layer.blocks):
            hooks.append(blk.register_forward_hook(
                lambda mod, a, o, k=f'layers.{i}.blocks.{j}': taps.__setitem__(k, o)))
    with torch.no_grad():
        y = bb(x)
        for k, v in taps.items():
            pack(out, 'swin.clean.' + k, v)
        pack(out, 'swin.clean.out', y)
        taps.clear()
        ym, _ = bb(x.clone(), vm)
        pack(out, 'swin.masked.out', ym)
    for h in hooks:
        h.remove()
    b3 = cf.cf_batch(3, tag='mid.bf')
    ids, mask = b3['token_ids'][:, 0], b3['input_mask'][:, 0]
    with torch.no_grad():
        t = m.text_backbone(ids, mask)['last_hidden_state']
        full(out, 'bert.last_hidden_state', t)
        vt = cf.cf_float('mid.bf.vt', (3, 2, 196, 192), 1.0)
        f = m.multimodal_backbone(visual_token=vt, text_input_mask=mask, text_input_embeds=t)
        full(out, 'fuse.t_last_hidden_state', f['t_last_hidden_state'])
        pack(out, 'fuse.v_last_hidden_state', f['v_last_hidden_state'])
    for B in [2, 4]:
        batch = cf.cf_batch(B, tag=f'mid.step{B}')
        m.zero_grad()
        loss, lv = _step(m, batch)
        loss.backward()
        for k, v in lv.items():
            out[f'B{B}.{k}'] = np.float64(v)
        named = dict(m.named_parameters())
        for k in MID_GRAD_KEYS:
            pack(out, f'B{B}.grad.{k}', named[k].grad)
        out[f'B{B}.n_unused'] = np.int64(sum(p.grad is None for p in named.values()))
        print('mid', B, {k: round(float(v), 5) for k, v in lv.items()})
    save('g_mid.npz', out)


def gen_test():
    """forward_test(separate_test=True) of the reference (:194-218): video / text embeddings for retrieval."""
    m, _ = ref_model()
    m.eval()
    H.init_dist_single()
    out = {}
    for B in [1, 3]:
        batch = cf.cf_batch(B, tag=f'test{B}')
        with torch.no_grad():
            v, t = m.forward_test(batch['imgs'], token_ids=batch['token_ids'], segment_ids=batch['segment_ids'],
                                  input_mask=batch['input_mask'])
        full(out, f'B{B}.visual_emb', v)
        full(out, f'B{B}.text_emb', t)
    save('g_test.npz', out)


FT_GRAD_KEYS = ['backbone.patch_embed.proj.weight', 'backbone.layers.0.blocks.1.attn.relative_position_bias_table',
                'backbone.layers.1.blocks.1.attn.qkv.weight', 'text_backbone.bert.embeddings.word_embeddings.weight',
                'text_backbone.bert.encoder.layer.1.attention.self.query.weight', 'ssl_head.img_projector.0.weight',
                'ssl_head.text_projector.0.weight']


def gen_finetune():
    """the retrieval fine-tuning path of the reference —
    NormSoftmaxLoss (losses/contrastive_loss.py:26-68), recall_for_video_text_retrieval
    (core/evaluation/accuracy.py:430-462) and CloverFinetune(task='retrieval')
    (recognizers/multimodal_transformer_finetune.py:59-86, :128-148)."""
    H.init_dist_single()
    m_cfg = cf.tiny_finetune_cfg()
    H.make_bert_dir(SCRATCH, hidden=cf.TINY_BERT['hidden_size'], layers=cf.TINY_BERT['num_hidden_layers'],
                    heads=cf.TINY_BERT['num_attention_heads'], inter=cf.TINY_BERT['intermediate_size'],
                    vocab=cf.TINY_BERT['vocab_size'], max_pos=cf.TINY_BERT['max_position_embeddings'])
    m = H.build_reference_model(m_cfg, SCRATCH)
    manifest = {k: list(v.shape) for k, v in m.state_dict().items()}
    missing, unexpected = m.load_state_dict(cf.cf_state(manifest), strict=False)
    assert all('relative_position_index' in k for k in missing) and not unexpected
    m.eval()
    out = {}

    # ---- the loss alone, on closed-form embeddings (values + input gradients)
    from mmaction.models.losses.contrastive_loss import NormSoftmaxLoss
    for cos in (False, True):
        for G in (1, 2, 4, 8, 33):
            for Dm in (128, 50):
                tag = f'loss.cos{int(cos)}.G{G}.D{Dm}'
                v = cf.cf_float(tag + '.v', (G, Dm), 1.3).requires_grad_()
                t = cf.cf_float(tag + '.t', (G, Dm), 0.9, 0.1).requires_grad_()
                loss = NormSoftmaxLoss(temperature=0.07 if not cos else 0.05, cos_sim=cos)(v, t)
                loss.backward()
                out[tag] = np.float64(loss.item())
                full(out, tag + '.dv', v.grad)
                full(out, tag + '.dt', t.grad)
    # zero-norm row: F.normalize / sim_matrix clamp the norm instead of dividing by zero
    for cos in (False, True):
        v = cf.cf_float('loss.zero.v', (4, 32), 1.0)
        v[2] = 0
        v.requires_grad_()
        t = cf.cf_float('loss.zero.t', (4, 32), 1.0).requires_grad_()
        loss = NormSoftmaxLoss(temperature=0.07, cos_sim=cos)(v, t)
        loss.backward()
        out[f'loss.zero.cos{int(cos)}'] = np.float64(loss.item())
        full(out, f'loss.zero.cos{int(cos)}.dt', t.grad)
    x = cf.cf_float('loss.sim', (6, 6), 9.0).requires_grad_()
    loss = NormSoftmaxLoss()(sim_mat=x)
    loss.backward()
    out['loss.sim'] = np.float64(loss.item())
    full(out, 'loss.sim.dx', x.grad)

    # ---- retrieval metrics
    from mmaction.core.evaluation.accuracy import recall_for_video_text_retrieval
    for N, D in ((1, 8), (7, 16), (50, 32), (200, 64)):
        ve = cf.cf_float(f'recall.N{N}.v', (N, D), 1.0).numpy()
        te = (0.35 * ve + cf.cf_float(f'recall.N{N}.t', (N, D), 1.0).numpy()).astype(np.float32)
        if N == 7:
            te[3] = 0                       # zero row: normalize_fn leaves it untouched
        r = recall_for_video_text_retrieval(ve, te)
        out[f'recall.N{N}'] = np.array([r['Recall@1'], r['Recall@5'], r['Recall@10'], r['MR'], r['Recall@all']],
                                       dtype=np.float64)
    sc = cf.cf_float('recall.scores', (12, 12), 1.0).numpy()
    r = recall_for_video_text_retrieval(input_scores=sc)
    out['recall.scores'] = np.array([r['Recall@1'], r['Recall@5'], r['Recall@10'], r['MR'], r['Recall@all']])

    # ---- the model: training step (loss + gradients) and separate_test inference (one clip and two clips/sample)
    aux = ['token_ids', 'segment_ids', 'input_mask']
    for B in (2, 4):
        batch = cf.cf_batch(B, tag=f'ft{B}')
        m.zero_grad()
        losses = m(batch['imgs'], batch['label'], return_loss=True, **{k: batch[k] for k in aux})
        loss, lv = m._parse_losses(losses)
        loss.backward()
        for k, v in lv.items():
            out[f'train.B{B}.{k}'] = np.float64(v)
        named = dict(m.named_parameters())
        for k in FT_GRAD_KEYS:
            pack(out, f'train.B{B}.grad.{k}', named[k].grad)
        out[f'train.B{B}.n_unused'] = np.int64(sum(p.grad is None for p in named.values()))
    batch = cf.cf_batch(4, tag='ft_test')
    with torch.no_grad():
        v, t = m.forward_test(batch['imgs'], **{k: batch[k] for k in aux})
        full(out, 'test.clips1.visual_emb', v)
        full(out, 'test.clips1.text_emb', t)
        imgs2 = batch['imgs'].reshape((2, 2) + tuple(batch['imgs'].shape[2:]))
        v, t = m.forward_test(imgs2, **{k: batch[k][:2] for k in aux})
        full(out, 'test.clips2.visual_emb', v)
        full(out, 'test.clips2.text_emb', t)
    save('g_finetune.npz', out)


def _dist_worker(rank, W, port, ret):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=W)
    m, _ = ref_model()
    m.eval()
    ddp = DDP(m, broadcast_buffers=False, find_unused_parameters=True)
    G = 4
    batch = cf.cf_batch(G, tag='dist')
    per = G // W
    shard = {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
    aux = {k: shard[k] for k in cf.AUX}
    losses = ddp(shard['imgs'], shard['label'], return_loss=True, **aux)
    loss, lv = m._parse_losses(losses)
    loss.backward()
    if rank == 0:
        named = dict(m.named_parameters())
        res = {'log_vars': lv, 'grads': {k: named[k].grad.detach().numpy().copy() for k in GRAD_KEYS}}
        ret.put(res)
    dist.barrier()
    dist.destroy_process_group()


def gen_dist():
    import torch.multiprocessing as mp
    out = {}
    ctx = mp.get_context('spawn')
    for W in [1, 2, 4]:
        q = ctx.Queue()
        procs = [ctx.Process(target=_dist_worker, args=(r, W, 29600 + W, q)) for r in range(W)]
        for p in procs:
            p.start()
        res = q.get(timeout=600)
        for p in procs:
            p.join()
        for k, v in res['log_vars'].items():
            out[f'W{W}.{k}'] = np.float64(v)
        for k, g in res['grads'].items():
            pack(out, f'W{W}.grad.{k}', torch.from_numpy(g))
        print(W, res['log_vars'])
    save('g_dist.npz', out)


# --------------------------------------------------------------------------- #
# BASELINE configs 2 and 5 at FULL shapes: the real reference on the weights / batch the GPU tests use
FULL_GRAD_KEYS = {
    'T': ['backbone.patch_embed.proj.weight', 'backbone.layers.2.blocks.3.attn.relative_position_bias_table',
          'backbone.layers.1.downsample.reduction.weight', 'backbone.layers.3.blocks.1.mlp.fc2.weight',
          'text_backbone.bert.encoder.layer.6.attention.self.query.weight',
          'multimodal_backbone.bert_encoder.layer.2.output.dense.weight', 'mlm_head.predictions.decoder.weight'],
    'B': ['backbone.patch_embed.proj.weight', 'backbone.layers.2.blocks.11.attn.relative_position_bias_table',
          'backbone.layers.2.blocks.14.mlp.fc1.weight', 'backbone.layers.2.blocks.17.attn.qkv.weight',
          'backbone.layers.1.downsample.reduction.weight', 'backbone.layers.3.blocks.1.mlp.fc2.weight',
          'multimodal_backbone.fc_in.weight', 'multimodal_backbone.bert_encoder.layer.2.output.dense.weight',
          'text_backbone.bert.encoder.layer.6.attention.self.query.weight', 'mlm_head.predictions.decoder.weight'],
}


def gen_full(cases=(('T', 8), ('B', 32)), B=2, fname='g_full.npz'):
    """The REAL reference at benchmark shapes: VideoSwin-T + BERT-base, 8 frames (BASELINE config 2)
    and VideoSwin-B + BERT-base, 32 frames (config 5's model and clip length), 224^2, 32 tokens, B = 2 — on exactly the
    weights (seed-4321 init of the registered modules) and batch (bench.synthetic_batch seed 77) that
    tests/gutil.py::full_size_oracle hands to the GPU tests.  Stored: the six logged losses, packed gradients of the
    parameters the full-size GPU tests look at, seconds per iteration of the reference on this container's cores."""
    import time
    ROOT = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, ROOT)
    import bench
    import clover_amd
    H.make_bert_dir(SCRATCH + '_base', hidden=768, layers=12, heads=12, inter=3072, vocab=30522, max_pos=512)
    H.init_dist_single()
    out = {}
    for variant, frames in cases:
        tag = f'{variant}{frames}'
        cfg = bench.model_cfg(variant, frames)
        torch.manual_seed(4321)
        own = clover_amd.build_model(cfg).eval()
        sd = {k: v.detach().clone() for k, v in own.state_dict().items()}
        del own
        rcfg = bench.model_cfg(variant, frames)
        for part in ('mm_backbone', 'text_backbone'):           # the reference classes read the HF directory instead
            rcfg[part].pop('bert_config', None)
        m = H.build_reference_model(rcfg, SCRATCH + '_base')
        missing, unexpected = m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()}, strict=False)
        assert all('relative_position_index' in k or 'position_ids' in k for k in missing), missing[:5]
        assert not unexpected, unexpected[:5]
        m.eval()
        batch = bench.synthetic_batch(B, frames, 32, seed=77)
        torch.set_num_threads(max(1, len(os.sched_getaffinity(0))))
        t0 = time.time()
        loss, lv = _step(m, batch)
        loss.backward()
        out[f'{tag}.ref_seconds_fwd_bwd'] = np.float64(time.time() - t0)
        out[f'{tag}.cores'] = np.int64(len(os.sched_getaffinity(0)))
        for k, v in lv.items():
            out[f'{tag}.{k}'] = np.float64(v)
        named = dict(m.named_parameters())
        for k in FULL_GRAD_KEYS[variant]:
            pack(out, f'{tag}.grad.{k}', named[k].grad)
        print(tag, {k: round(float(v), 6) for k, v in lv.items()}, 'seconds', out[f'{tag}.ref_seconds_fwd_bwd'])
        del m, named, loss
    save(fname, out)


def gen_full8():
    """BASELINE config 2 at the BENCHMARK's batch: VideoSwin-T + BERT-base, 8 clips x 8 frames x 224^2, 32 tokens — the shapes
    bench.py times (tests/test_step_gpu.py::test_bench_shapes_step_matches_reference)."""
    gen_full(cases=(('T', 8),), B=8, fname='g_full_b8.npz')


def gen_scaler():
    """The reference's LossScaler (core/hooks/fp16_utils.py:285-389) driven by fixed overflow sequences: the scale BEFORE
    every iteration and the scaler's state_dict at the end."""
    H.install_shims()
    from mmaction.core.hooks.fp16_utils import LossScaler
    out = {}
    for name, kw, flags in cf.scaler_cases():
        sc = LossScaler(**kw)
        scales = []
        for f in flags:
            scales.append(float(sc.loss_scale))
            sc.update_scale(bool(f))
        st = sc.state_dict()
        out[name + '.flags'] = np.array(flags, dtype=np.int8)
        out[name + '.scales'] = np.array(scales, dtype=np.float64)
        out[name + '.final'] = np.array([st['cur_scale'], st['cur_iter'], st['last_overflow_iter'], st['scale_factor'],
                                         st['scale_window']], dtype=np.float64)
    save('g_scaler.npz', out)


SETS = dict(scaler=gen_scaler, idx=gen_idx, swin=gen_swin, bert_fuse=gen_bert_fuse, heads_loss=gen_heads_loss,
            step=gen_step, mid=gen_mid, dist=gen_dist, inflate=gen_inflate, test=gen_test, finetune=gen_finetune)

HEAVY = dict(full=gen_full, full8=gen_full8)      # only on request

if __name__ == '__main__':
    which = sys.argv[1:] or list(SETS)
    SETS.update(HEAVY)
    torch.set_num_threads(8)
    for s in which:
        SETS[s]()
