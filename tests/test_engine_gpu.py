"""GPU tests of the training engine: flat-slab AdamW + clip vs torch.optim on the same gradients,
eager vs hipGraph-captured steps, loss decrease.  `-m gpu` only."""
import copy

import pytest
import torch

import closed_form as cf
import gutil

pytestmark = pytest.mark.gpu
DEV = 'cuda'
from clover_amd import _lib as _clv_lib  # noqa: E402
HALF = _clv_lib.half_dtype()          # the library's 16-bit element type


def make_model():
    import clover_amd
    m = clover_amd.build_model(cf.tiny_model_cfg())
    m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False)
    return m.to(DEV).eval()          # eval: dropout off -> deterministic trajectories


def batch(B=2, tag='eng'):
    return {k: v.to(DEV) for k, v in cf.cf_batch(B, tag=tag).items()}


def test_engine_matches_torch_adamw_and_clip():
    from clover_amd.engine import CloverEngine, paramwise_weight_decay
    b = batch()
    m1, m2 = make_model(), make_model()
    eng = CloverEngine(m1, b, lr=1e-3, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
    assert sorted(eng.unused_names) == gutil.unused_params()
    wd = paramwise_weight_decay(m2, 0.005, 0.0, 0.0, {'relative_position_bias_table': dict(decay_mult=0.)})
    named = [(n, p) for n, p in m2.named_parameters() if n not in eng.unused_names]
    opt = torch.optim.AdamW([dict(params=[p], weight_decay=wd[n]) for n, p in named], lr=1e-3, betas=(0.9, 0.98),
                            eps=1e-8)
    for it in range(3):
        eng.step(b)
        opt.zero_grad(set_to_none=True)
        m2.train_step(b, None)['loss'].backward()
        torch.nn.utils.clip_grad_norm_([p for _, p in named], 15.0)
        for g in opt.param_groups:
            g['lr'] = 1e-3 * (1e-3 + 0.5 * (1 - 1e-3) * (1 + 1.0))      # cosine_lr at it << max_iters == base lr
        opt.step()
    p1 = dict(m1.named_parameters())
    worst = 0.0
    for n, p in named:
        d = (p1[n].detach() - p.detach()).abs().max().item()
        worst = max(worst, d / (p.detach().abs().max().item() + 1e-6))
    # identical math on bf16-noisy gradients: Adam's sign-like update makes tiny grads flip — the key part of a q|k|v
    # bias has a mathematically ZERO gradient (softmax is invariant to it), so its update is pure rounding noise and two
    # correct implementations may move it in opposite directions: 2 * 3 steps * lr = 6e-3 on parameters of size 0.05
    assert worst < 0.13, worst


def test_engine_lr_mult_and_fractional_decay_match_torch_param_groups():
    """any decay_mult / lr_mult of a paramwise_cfg gets its own slab (mmcv makes one param group per
    parameter); compared with torch.optim.AdamW over the same per-parameter options."""
    from clover_amd.engine import CloverEngine, paramwise_options
    b = batch(tag='pw')
    pw = dict(norm_decay_mult=0.0, bias_decay_mult=0.0, bias_lr_mult=2.0,
              custom_keys={'backbone.layers': dict(lr_mult=0.25, decay_mult=0.5),
                           'relative_position_bias_table': dict(decay_mult=0.)})
    m1, m2 = make_model(), make_model()
    eng = CloverEngine(m1, b, lr=1e-3, weight_decay=0.01, paramwise_cfg=pw, grad_clip=15.0, max_iters=10 ** 9)
    assert len(eng.segments) >= 4 and len({(sg.weight_decay, sg.lr_mult) for sg in eng.segments}) == len(eng.segments)
    o = paramwise_options(m2, 0.01, pw)
    named = [(n, p) for n, p in m2.named_parameters() if n not in eng.unused_names]
    opt = torch.optim.AdamW([dict(params=[p], weight_decay=o[n][0], lr=1e-3 * o[n][1]) for n, p in named],
                            lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
    for it in range(3):
        eng.step(b)
        opt.zero_grad(set_to_none=True)
        m2.train_step(b, None)['loss'].backward()
        torch.nn.utils.clip_grad_norm_([p for _, p in named], 15.0)
        opt.step()
    p1 = dict(m1.named_parameters())
    p0 = dict(make_model().named_parameters())
    for sg in eng.segments:                              # every parameter sits in the slab of ITS options
        for n in sg.names:
            assert o[n] == (sg.weight_decay, sg.lr_mult), (n, o[n], sg.weight_decay, sg.lr_mult)
    for n, p in named:
        # Adam moves each element by ~lr * lr_mult per step whatever the gradient's size (its sign may be noise):
        # compare the mean displacement — a wrong lr_mult shows as a 2-8x ratio
        mine = (p1[n].detach() - p0[n]).abs().mean().item()
        ref = (p.detach() - p0[n]).abs().mean().item()
        # a key bias shifts every score of a softmax row equally: its exact gradient is zero, what Adam normalises
        # there is rounding noise of either implementation, and the displacement ratio is not a statement about lr_mult
        hi = 2.5 if n.endswith('key.bias') else 1.6
        assert 0.6 * ref - 1e-9 <= mine <= hi * ref + 1e-9, (n, mine, ref, o[n])
        assert ref <= 3.3 * 1e-3 * o[n][1] + 1e-7, (n, ref, o[n])


def test_transposed_weight_shadows_follow_the_optimizer():
    """The engine keeps bf16 W^T copies of the Linear weights whose input gradient runs on clv_gemm_nt; they must equal
    the transposed bf16 shadow after construction, after every optimizer step and after a checkpoint load."""
    from clover_amd.engine import CloverEngine
    b = batch(tag='wt')
    eng = CloverEngine(make_model(), b, lr=1e-3, weight_decay=0.005, grad_clip=15.0, max_iters=100)
    # members of a fused view (BERT Q|K|V, one [3H, H] GEMM) are served by the transpose of the FUSED matrix
    fused = [f for sg in eng.segments for f in sg._fused if getattr(f, '_clv_want_t', False)]
    in_fused = {id(m) for f in fused for m in f._clv_members}
    flagged = [p for sg in eng.segments for p in sg.params if getattr(p, '_clv_want_t', False) and id(p) not in in_fused]
    flagged += fused
    assert flagged and fused and all(hasattr(p, '_clv_shadow_t') for p in flagged)

    def check():
        for p in flagged:
            assert p._clv_shadow_t.shape == (p.shape[1], p.shape[0])
            assert torch.equal(p._clv_shadow_t, p._clv_shadow.t().contiguous())
            assert torch.equal(p._clv_shadow, p.data.to(p._clv_shadow.dtype))
    check()
    w0 = flagged[0]._clv_shadow_t.clone()
    eng.step(b)
    check()
    assert not torch.equal(w0, flagged[0]._clv_shadow_t)
    with torch.no_grad():
        for p in flagged:
            p.mul_(0.5)
    eng.refresh_shadow()
    check()


def test_skipped_step_keeps_adam_count_and_dry_step_has_no_side_effects():
    """A non-finite gradient norm skips the update on the device and does not advance Adam's step count (the
    reference skips optimizer.step(), mmcv_Fp16OptimizerHook.py:123-141) while the LR index moves on; dry_step —
    the warm-up tools/train.py runs before capturing graphs — changes nothing."""
    from clover_amd.engine import CloverEngine
    b = batch(tag='skip')
    eng = CloverEngine(make_model(), b, lr=1e-3, weight_decay=0.005, grad_clip=15.0, max_iters=100, warmup_iters=10)
    snap = [(sg.flat_p.clone(), sg.exp_avg.clone(), sg.exp_avg_sq.clone(), sg.shadow.clone()) for sg in eng.segments]
    eng.dry_step(b)
    for sg, (p0, m0, v0, s0) in zip(eng.segments, snap):
        assert torch.equal(sg.flat_p, p0) and torch.equal(sg.exp_avg, m0) and torch.equal(sg.exp_avg_sq, v0)
        assert torch.equal(sg.shadow, s0) and float(sg.flat_g.abs().max()) == 0.0
    assert eng.step_count == 0 and eng.lr_iter == 0 and eng.adam_steps() == 0
    lr0 = eng.current_lr()
    eng.step(b)
    assert eng.last_lr == lr0 and eng.adam_steps() == 1                     # the first step uses schedule index 0
    snap = [(sg.flat_p.clone(), sg.exp_avg.clone()) for sg in eng.segments]
    eng.model.train_step(b, None)['loss'].backward()
    # (a slot the norm pass reads: with the fused norm the first-touch weight gradients deliver their sums from the kernels
    # that write them, so a value poked into one of THOSE afterwards is invisible — as it should be)
    (eng._zero_views[0] if eng._zero_views else eng.segments[0].flat_g)[7] = float('nan')
    eng.reducer.finish()
    eng.optimizer_step()
    for sg, (p0, m0) in zip(eng.segments, snap):
        assert torch.equal(sg.flat_p, p0) and torch.equal(sg.exp_avg, m0)
    # gradients are cleared all the same: what accumulates is zero again, the first-touch slots are marked stale
    for v in (eng._zero_views if eng._zero_views is not None else [sg.flat_g for sg in eng.segments]):
        assert float(v.abs().nan_to_num(0).max()) == 0.0
    assert not eng._ft.done
    st = eng.optimizer_state()
    assert st['step'] == 1 and st['skipped'] == 1 and st['calls'] == 2 and eng.lr_iter == 2
    eng.step(b)
    assert eng.adam_steps() == 2
    # bias corrections after the skipped step are those of t = 2
    from clover_amd import ops
    s = ops.optim_state_read(eng.optim_state)
    assert abs(s['bc1'] - (1 - 0.9 ** 2)) < 1e-6 and abs(s['bc2_sqrt'] - (1 - 0.98 ** 2) ** 0.5) < 1e-6
    assert s['norm'] > 0 and s['coef'] <= 1.0


@pytest.mark.parametrize('mode', ['eager', 'graph'])
def test_dynamic_loss_scaler_on_the_device_follows_the_reference_rule(mode):
    """`fp16 = dict(loss_scale='dynamic')` (pretrain_webvid_cc3m.py:21): the engine's loss scaler lives on the device and is
    moved by clv_optim_prep.  Started far too high (every step overflows until the scale fits), with a short growth window:
    the scale before every step must equal the reference's LossScaler (oracle/loss_scaler.py, pinned on the reference class)
    driven by the overflow flags the device reports; a skipped step leaves parameters and Adam's count alone; a captured
    step follows the moving scale without re-capture; taken steps train (loss falls); the scaler state round-trips through
    optimizer_state() like runner.meta['fp16']['loss_scaler'] (mmcv_Fp16OptimizerHook.py:74-78,147-149)."""
    from clover_amd import ops
    from clover_amd.engine import CloverEngine
    from oracle.loss_scaler import LossScaler
    b = batch(2, 'dyn')
    kw = dict(init_scale=2.0 ** 120, mode='dynamic', scale_factor=2.0 ** 12, scale_window=3)
    eng = CloverEngine(make_model(), b, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9, loss_scale=kw)
    if mode == 'graph':
        assert eng.capture(b)
    ref = LossScaler(**kw)
    assert eng.loss_scale == ref.loss_scale
    taken, skipped, losses = 0, 0, []
    for it in range(26):
        p0 = [sg.flat_p.clone() for sg in eng.segments]
        out = eng.step(b)
        st = ops.optim_state_read(eng.optim_state)
        ref.update_scale(bool(st['skip']))
        assert st['loss_scale'] == ref.loss_scale and st['scale_iter'] == ref.cur_iter, (it, st, ref.state_dict())
        assert st['last_overflow'] == ref.last_overflow_iter
        same = all(torch.equal(sg.flat_p, q) for sg, q in zip(eng.segments, p0))
        assert same == bool(st['skip']), (it, st)
        taken += 0 if st['skip'] else 1
        skipped += st['skip']
        if not st['skip']:
            losses.append(float(out['log_vars']['loss']))
        assert st['t'] == taken and st['skipped'] == skipped
    print(mode, 'taken', taken, 'skipped', skipped, 'scale', eng.loss_scale, losses[:2], losses[-2:])
    assert skipped >= 5 and taken >= 8                        # 2**120 needs >= 9 divisions by 2**12 before anything fits
    assert losses[-1] < losses[0]
    # the logged loss is the true (unscaled) one, finite even on skipped steps
    assert all(l == l and abs(l) < 1e4 for l in losses)
    state = eng.optimizer_state()
    assert state['loss_scaler'] == dict(ref.state_dict(), cur_scale=float(ref.cur_scale))
    eng2 = CloverEngine(make_model(), b, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9, loss_scale=kw)
    eng2.load_optimizer_state(state)
    assert eng2.loss_scaler_state() == state['loss_scaler'] and eng2.adam_steps() == taken


def test_static_loss_scale_equals_unscaled_update_in_exact_arithmetic():
    """A static power-of-two scale is exact wherever nothing under- or overflows: two engines that differ only in the scale
    (256 and 4096) must take (nearly) the same first step — the scale is multiplied in at the root and divided out in
    clv_optim_prep."""
    from clover_amd.engine import CloverEngine
    b = batch(2, 'stat')
    ps = []
    for scale in (256.0, 4096.0):
        eng = CloverEngine(make_model(), b, lr=1e-3, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9, loss_scale=scale)
        assert eng.loss_scale == scale
        eng.step(b)
        ps.append((torch.cat([sg.flat_p for sg in eng.segments]).clone(), eng.grad_norm()))
    (p1, n1), (p2, n2) = ps
    assert abs(n1 - n2) <= 2e-3 * n1, (n1, n2)
    # Adam's first step is lr * sign(g): only gradients at rounding-noise level may differ
    assert float(((p1 - p2).abs() > 1e-4).float().mean()) < 0.02
    m = make_model()
    eng = CloverEngine(m, b, lr=1e-3, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9, loss_scale=1.0)
    assert eng.loss_scale == 1.0 and m._clv_loss_scale_dev is False and eng.loss_scaler_state() is None
    eng.step(b)
    assert eng.adam_steps() == 1


@pytest.mark.parametrize('mode', ['eager', 'graph'])
def test_fused_gradient_norm_equals_the_norm_of_the_slabs(mode, monkeypatch):
    """One-rank engines take the norm's sum of squares of the first-touch weight gradients from the kernels that write them
    (clv_linear_wgrad_batch_ss / clv_wgrad_fold_batch_ss norm slots) and read only the rest of the slabs
    (clv_sumsq_ranges): the norm the optimizer used must equal the norm of what is in the slabs — computed here from a copy of
    the gradients taken before the optimizer step — and the engine with CLOVER_FUSED_NORM=0; also after a dry step and a
    skipped backward (stale partial sums must not leak into the next step)."""
    from clover_amd import ops
    from clover_amd.engine import CloverEngine
    b = batch(2, 'norm')
    eng = CloverEngine(make_model(), b, lr=1e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
    assert eng._norm_tables is not None and eng.first_touch_params > 0.5 * eng.num_params
    covered = sum(int(t[:n, 1].sum()) for t, n in eng._norm_tables)
    assert covered + eng.first_touch_params == sum(sg.flat_g.numel() for sg in eng.segments)
    eng.dry_step(b)                                          # leaves no partial sums behind
    assert float(eng.sumsq.abs().max()) == 0.0
    if mode == 'graph':
        assert eng.capture(b)
        assert float(eng.sumsq.abs().max()) == 0.0
    for it in range(3):
        # the gradients of this step, copied between the backward and the optimizer
        eng.reducer.begin_step()
        if mode == 'graph':
            eng._graphed_forward_backward(b)
        else:
            eng._ft.done.clear()
            out = eng.model.train_step(b, None)
            eng._backward(lambda: out['loss'].backward())
        eng.finish_backward()
        ref = torch.cat([sg.flat_g.double().reshape(-1) for sg in eng.segments]).norm().item() / eng.loss_scale
        eng.optimizer_step()
        got = eng.grad_norm()
        assert abs(got - ref) <= 2e-5 * ref, (it, got, ref)
        assert float(eng.sumsq.abs().max()) == 0.0           # prep re-zeroed the accumulator and the slots
    monkeypatch.setenv('CLOVER_FUSED_NORM', '0')
    eng0 = CloverEngine(make_model(), b, lr=1e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
    assert eng0._norm_tables is None
    eng1 = CloverEngine(make_model(), b, lr=1e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
    monkeypatch.delenv('CLOVER_FUSED_NORM')
    eng0.step(b)
    n0 = eng0.grad_norm()
    eng2 = CloverEngine(make_model(), b, lr=1e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
    eng2.step(b)
    assert abs(eng2.grad_norm() - n0) <= 1e-4 * n0, (eng2.grad_norm(), n0)
    del eng1


def test_sumsq_ranges_kernel():
    """clv_sumsq_ranges against torch on ragged ranges (tails that are not multiples of 4, ranges longer than a chunk, an
    empty table)."""
    from clover_amd import ops
    g = torch.randn(300000, device=DEV)
    ranges = [(0, 7), (8, 8), (16, 40000), (40004, 40005), (100000, 300000)]
    tab, nblk = ops.sumsq_range_table(ranges, g.device)
    acc = torch.zeros(4, device=DEV)
    ops.sumsq_ranges(g, tab, nblk, acc)
    ref = sum(float((g[a:b].double() ** 2).sum()) for a, b in ranges)
    assert abs(float(acc[0]) - ref) <= 1e-5 * ref and float(acc[1:].abs().max()) == 0.0
    tab0, n0 = ops.sumsq_range_table([], g.device)
    ops.sumsq_ranges(g, tab0, n0, acc)
    assert abs(float(acc[0]) - ref) <= 1e-5 * ref


def test_engine_gradient_slab_equals_plain_autograd():
    """The engine's plumbing — gradient sinks into the flat fp32 slab, bf16 shadow weights, fused Q|K|V slab
    views (clv_fuse_groups) — must leave exactly the gradients plain autograd computes on an identical model."""
    from clover_amd.engine import CloverEngine
    from clover_amd.backbones.bert_layers import BertSelfAttention
    b = batch(tag='slab')
    m1, m2 = make_model(), make_model()
    eng = CloverEngine(m1, b, lr=1e-3, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
    fused = [m for m in m1.modules() if isinstance(m, BertSelfAttention) and getattr(m, '_clv_fused', None)]
    assert fused, 'no BERT self-attention got fused Q|K|V views'
    w, bb = fused[0]._clv_fused
    assert w.shape[0] == 3 * fused[0].query.weight.shape[0] and w.data_ptr() == fused[0].query.weight.data_ptr()
    assert bb._clv_grad.data_ptr() == fused[0].query.bias.grad.data_ptr()
    m1.train_step(b, None)['loss'].backward()
    m2.train_step(b, None)['loss'].backward()
    p1 = dict(m1.named_parameters())
    for n, p in m2.named_parameters():
        if n in eng.unused_names:
            continue
        ref = p.grad.detach().float()
        # (the engine's slabs hold the gradients times the loss scale of the 16-bit backward: 1024 in the fp16 build)
        err = (p1[n].grad.float() / eng.loss_scale - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        assert err < 2e-2, (n, err)


@pytest.mark.parametrize('loss_graph', ['0', '1'])         # the loss section eager / as a hipGraph of its own
def test_graph_capture_equals_eager_and_loss_decreases(loss_graph, monkeypatch):
    from clover_amd.engine import CloverEngine
    monkeypatch.setenv('CLOVER_LOSS_GRAPH', loss_graph)
    b = batch(4, 'eng4')
    traj = {}
    for mode in ('eager', 'graph'):
        m = make_model()
        eng = CloverEngine(m, b, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
        if mode == 'graph':
            eng.step(b)
            assert eng.capture(b)
            assert (eng.graph_loss is not None) == (loss_graph == '1')
            assert eng.input_buffers() is not None
            b_run = eng.input_buffers()                      # a loader writing straight into the static buffers
        else:
            eng.step(b)
            assert eng.input_buffers() is None
            b_run = b
        losses = []
        for _ in range(6):
            out = eng.step(b_run)
            losses.append(out['log_vars']['loss'])
        traj[mode] = losses
    print(traj)
    for a, g in zip(traj['eager'], traj['graph']):
        assert abs(a - g) < 0.05 * max(1.0, abs(a)), (traj['eager'], traj['graph'])
    assert traj['graph'][-1] < traj['graph'][0] - 0.5          # the step actually trains


def test_use_checkpoint_under_the_engine():
    """``use_checkpoint=True`` through the engine — dry_step, hipGraph capture (the recompute runs inside the
    captured backward: checkpoint must not touch the RNG state there), replayed steps, first-touch sinks.  Every
    parameter's gradient of one replayed forward / backward equals the non-checkpointed engine's on the same weights,
    and three optimizer steps give the same losses."""
    import clover_amd
    from clover_amd.engine import CloverEngine
    b = batch(2, 'ckeng')
    res = {}
    for ck in (False, True):
        cfg = cf.tiny_model_cfg()
        cfg['backbone']['use_checkpoint'] = ck
        m = clover_amd.build_model(cfg)
        m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False)
        m = m.to(DEV).eval()
        eng = CloverEngine(m, b, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
        eng.dry_step(b)
        assert eng.capture(b)
        eng.zero_grads()
        eng._graphed_forward_backward(b)
        torch.cuda.synchronize()
        g = {}
        for seg in eng.segments:
            for name, p_, off in zip(seg.names, seg.params, seg.offsets):
                g[name] = seg.flat_g[off:off + p_.numel()].clone()
        eng.zero_grads()
        losses = [float(eng.step(b)['log_vars']['loss']) for _ in range(3)]
        res[ck] = (g, losses, eng.first_touch_params)
    assert res[True][2] == res[False][2] and res[True][2] > 0          # the same slots qualify as first-touch
    for n, r in res[False][0].items():
        scale = float(r.abs().max())
        assert float((res[True][0][n] - r).abs().max()) <= 1e-3 * scale + 1e-9, n
    for a, c in zip(res[False][1], res[True][1]):
        assert abs(a - c) <= 1e-3 * max(1.0, abs(a)), res


def test_optimizer_state_from_another_slab_layout():
    """a checkpoint whose flat Adam moments were laid out with other slot offsets (the slab alignment / phantom
    padding changed between rounds) must load per parameter by name, not by copying the flat buffers."""
    from clover_amd.engine import CloverEngine
    b = batch(2, 'optlay')
    eng = CloverEngine(make_model(), b, lr=1e-3, weight_decay=0.01, grad_clip=15.0, max_iters=10 ** 9)
    eng.step(b)
    eng.step(b)
    st = eng.optimizer_state()
    for seg_state, sg in zip(st['segments'], eng.segments):
        off, pos, m_chunks, v_chunks = [], 0, [], []
        for i, p_ in enumerate(sg.params):
            n, pad = p_.numel(), (i % 3) * 2                  # the "old" layout: other gaps between the slots
            off.append(pos)
            for chunks, key in ((m_chunks, 'exp_avg'), (v_chunks, 'exp_avg_sq')):
                chunks.append(seg_state[key][sg.offsets[i]:sg.offsets[i] + n])
                chunks.append(torch.full((pad,), 7.0))         # garbage in the gaps must not reach any parameter
            pos += n + pad
        seg_state['offsets'] = off + [pos]
        seg_state['exp_avg'], seg_state['exp_avg_sq'] = torch.cat(m_chunks), torch.cat(v_chunks)
        assert seg_state['exp_avg'].numel() != sg.exp_avg.numel() or off != list(sg.offsets[:len(off)])
    eng2 = CloverEngine(make_model(), b, lr=1e-3, weight_decay=0.01, grad_clip=15.0, max_iters=10 ** 9)
    eng2.load_optimizer_state(st)
    assert eng2.adam_steps() == eng.adam_steps()
    for sg, sg2 in zip(eng.segments, eng2.segments):
        for i, (name, p_) in enumerate(zip(sg.names, sg.params)):
            a, n = sg.offsets[i], p_.numel()
            assert torch.equal(sg.exp_avg[a:a + n], sg2.exp_avg[a:a + n]), name
            assert torch.equal(sg.exp_avg_sq[a:a + n], sg2.exp_avg_sq[a:a + n]), name
        assert float(sg2.exp_avg.abs().max()) < 7.0


def test_graphs_per_batch_geometry():
    """The reference alternates video (many-frame) and image (1-frame, padded to 2) batches (clover_runner.py:76-93):
    the engine keeps one set of hipGraphs per batch geometry, captured on first sight, and the alternating
    trajectory equals the eager one."""
    from clover_amd.engine import CloverEngine
    vid = batch(2, 'geo_v')
    img = {k: v.to(DEV) for k, v in cf.cf_batch(2, frames=1, tag='geo_i').items()}
    traj = {}
    for mode in ('eager', 'graph'):
        eng = CloverEngine(make_model(), vid, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
        eng.step(vid)
        if mode == 'graph':
            assert eng.capture(vid)
        losses = []
        for _ in range(3):
            losses.append(eng.step(vid)['log_vars']['loss'])
            losses.append(eng.step(img)['log_vars']['loss'])
        traj[mode] = losses
        if mode == 'graph':
            assert len(eng._captures) == 2
            # the lazily captured second geometry must deliver EVERY parameter's gradient inside its graphs
            # (including the ones that go through autograd's AccumulateGrad rather than a kernel-side sink)
            for b in (img, vid):
                for seg in eng.segments:
                    seg.flat_g.zero_()
                if eng._signature(b) != eng._active_sig:
                    eng._activate(eng._signature(b))
                eng._graphed_forward_backward(b)
                torch.cuda.synchronize()
                got = [seg.flat_g.clone() for seg in eng.segments]
                for seg in eng.segments:
                    seg.flat_g.zero_()
                eng.model.train_step(b, None)['loss'].backward()
                torch.cuda.synchronize()
                for seg, g in zip(eng.segments, got):
                    for name, p_, off in zip(seg.names, seg.params, seg.offsets):
                        a, r = g[off:off + p_.numel()], seg.flat_g[off:off + p_.numel()]
                        scale = r.abs().max().item()
                        if scale > 0:
                            assert (a - r).abs().max().item() <= 3e-2 * scale + 1e-6, (name, scale)
                        else:
                            assert a.abs().max().item() == 0, name
                for seg in eng.segments:
                    seg.flat_g.zero_()
    print(traj)
    for a, g in zip(traj['eager'], traj['graph']):
        assert abs(a - g) < 0.05 * max(1.0, abs(a)), traj


def test_drop_path_tables_survive_a_second_batch_size():
    """in train mode every captured forward graph bakes the address of the DropPath keep-probability
    table its bernoulli launch reads.  A second batch geometry with ANOTHER per-rank B must not free / replace the table
    of the first one: tables are kept per (device, B) for the life of the module, the first geometry's graphs replay on
    the very same memory after the second capture, and its DropPath factors stay in {0, 1 / keep}."""
    import clover_amd
    from clover_amd.engine import CloverEngine
    m = clover_amd.build_model(cf.tiny_model_cfg(drop=0.2))
    m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False)
    m = m.to(DEV).train()
    b2, b3 = batch(2, 'dp2'), batch(3, 'dp3')
    eng = CloverEngine(m, b2, lr=1e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9)
    assert eng.capture(b2)
    bb = m.backbone
    keys0 = dict(bb._dp_tables)
    assert len(keys0) == 1                                   # the doubled clean + masked pass: one table for 2B samples
    (key0, (keep0, inv0)), = keys0.items()
    ptr0 = keep0.data_ptr()
    eng.step(b2)
    eng.step(b3)                                             # captures the second geometry (another B)
    assert len(eng._captures) == 2 and len(bb._dp_tables) == 2
    assert bb._dp_tables[key0][0] is keep0 and keep0.data_ptr() == ptr0          # not replaced, not freed
    expect = keep0.clone()
    for _ in range(3):
        lv = eng.step(b2)['log_vars']                        # replays the FIRST geometry's graphs
        assert all(v == v and abs(v) < 1e4 for v in (float(x) for x in lv.values())), dict(lv)
        eng.step(b3)
    torch.cuda.synchronize()
    assert torch.equal(keep0, expect) and float(keep0.min()) > 0.5        # the table the first graphs read is intact


def make_finetune_model():
    import clover_amd
    m = clover_amd.build_model(cf.tiny_finetune_cfg())
    sd = {k: v for k, v in cf.cf_state(gutil.manifest()).items() if k in m.state_dict()}
    m.load_state_dict(sd, strict=False)
    return m.to(DEV).eval()


def test_finetune_engine_graph_equals_eager_and_trains(monkeypatch):
    """the retrieval fine-tuning recognizer through the same engine — eager steps, hipGraph steps
    (no rank-local loss: the backward graph has one root), and the 1-rank RCCL path with the video-encoder cut."""
    import os
    import torch.distributed as dist
    from clover_amd.engine import CloverEngine
    b = batch(4, 'fteng')

    def run(mode):
        m = make_finetune_model()
        eng = CloverEngine(m, b, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9, bucket_mb=1)
        # the fusion encoder (built, never run by the retrieval task), the mask token, BERT's pooler: as in the reference
        assert len(eng.unused_names) == int(gutil.load('g_finetune.npz')['train.B4.n_unused'])
        assert any(n.startswith('multimodal_backbone.') for n in eng.unused_names)
        eng.step(b)
        if mode != 'eager':
            assert eng.capture(b)
        return [eng.step(b)['log_vars']['loss'] for _ in range(6)], eng

    eager, _ = run('eager')
    graph, eng = run('graph')
    assert eng.graph_bwd_video is None
    monkeypatch.setenv('CLOVER_FORCE_COLLECTIVES', '1')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29563')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        rccl, eng = run('rccl')
        assert eng.reducer.active and eng.graph_bwd_video is not None and eng.graph_bwd_text is not None
    finally:
        dist.destroy_process_group()
    print(eager, graph, rccl)
    for a, g, r in zip(eager, graph, rccl):
        assert abs(a - g) < 0.05 * max(1.0, abs(a)) and abs(a - r) < 0.05 * max(1.0, abs(a)), (eager, graph, rccl)
    assert graph[-1] < graph[0] - 0.2


def test_rccl_code_path_on_one_gpu(monkeypatch):
    """The N>1 code path (packed RCCL all-gather with local-slice backward, logged-scalar all-reduce,
    bucketed gradient all-reduce on the side stream, hipGraph mode) on a real 1-rank RCCL group:
    must give the same trajectory as the plain single-process path."""
    import os
    import torch.distributed as dist
    from clover_amd.engine import CloverEngine
    b = batch(2, 'rccl')

    def run():
        m = make_model()
        m.backbone.CUT_STAGE = 0          # the tiny 2-stage encoder has no stage 2: put the in-encoder cut at stage 0
        eng = CloverEngine(m, b, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10 ** 9, bucket_mb=1)
        eng.step(b)
        eng.capture(b)
        return [eng.step(b)['log_vars']['loss'] for _ in range(3)], eng

    def grads(eng, graph=True):
        """every parameter's gradient of one more forward/backward (+ collectives), by name"""
        for seg in eng.segments:
            seg.flat_g.zero_()
        if graph:
            eng._graphed_forward_backward(b)
        else:
            eng.model.train_step(b, None)['loss'].backward()
        eng.reducer.finish()
        torch.cuda.synchronize()
        out = {}
        for seg in eng.segments:
            for name, p_, off in zip(seg.names, seg.params, seg.offsets):
                out[name] = seg.flat_g[off:off + p_.numel()].clone()
        return out

    ref, _ = run()
    monkeypatch.setenv('CLOVER_FORCE_COLLECTIVES', '1')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29561')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        got, eng = run()
        assert eng.reducer.active and len(eng.reducer.buckets) > 1
        assert len(eng.graph_bwd_video) == 2 and eng.graph_bwd_text is not None        # four backward graphs
        classes = {eng._pclass.get(id(q), 'h') for ps in eng.reducer._bucket_params for q in ps}
        assert classes == {'h', 't', 'v1', 'v0'}
        assert all(len({eng._pclass.get(id(q), 'h') for q in ps}) == 1 for ps in eng.reducer._bucket_params)
        # bucket launch order of the last step: heads / fusion first (right after graph 1), text-encoder and late-video
        # buckets from between the replays, and only the early video stages (+ patch embedding) are left to finish();
        # the gradients travel as bf16 (2 bytes per element on the wire)
        log = list(eng.reducer.launch_log)
        assert eng.wire is not None and len(log) == len(eng.reducer.buckets)
        order = [k for _, k, _, _ in log]
        nh = order.count('h')
        assert nh > 0 and order[:nh] == ['h'] * nh and order[-1] == 'v0'
        assert {k for _, k, _, ph in log if ph == 'finish'} == {'v0'} and {ph for _, k, _, ph in log if k != 'v0'} == {'where'}
        total, left = sum(b for _, _, b, _ in log), sum(b for _, _, b, ph in log if ph == 'finish')
        assert total == 2 * sum(seg.flat_g.numel() for seg in eng.segments) and left <= 0.05 * total, (left, total)
        # every bucket's bf16 wire copy is written inside the backward graph that completes it (no eager pack launches),
        # and the stall of the compute stream on the comm stream is recorded
        assert eng._prepacked == frozenset(range(len(eng.reducer.buckets))), eng._prepacked
        eng.reducer.start_timing()
        eng.step(b)
        torch.cuda.synchronize()
        ex = eng.reducer.exposed_ms()
        assert ex is not None and ex >= 0.0
        g_got, g_ref = grads(eng), grads(eng, graph=False)
    finally:
        dist.destroy_process_group()
    for a, g in zip(ref, got):
        assert abs(a - g) < 0.02 * max(1.0, abs(a)), (ref, got)
    # the cut backward (heads | text encoder || video encoder, buckets leaving in between) delivers every
    # parameter's gradient, equal to the plain eager backward's on the same weights
    assert set(g_got) == set(g_ref)
    for name, r in g_ref.items():
        scale = r.abs().max().item()
        assert (g_got[name] - r).abs().max().item() <= 3e-2 * scale + 1e-6, name


def test_checkpoint_roundtrip_through_engine(tmp_path):
    """save from a trained engine, `load_from` into a fresh one (weights only) and `resume` into another
    (weights + AdamW moments + counters): the loaded engines must compute from the loaded weights (the bf16 compute
    copy is re-derived), and the resumed one must continue the trajectory; a pre-training checkpoint loads into
    the fine-tuning recognizer with only the pre-training heads left over."""
    from clover_amd.engine import CloverEngine
    from clover_amd.runner import CloverRunner
    b = batch(2, 'ckpt')
    kw = dict(lr=2e-4, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
    e1 = CloverEngine(make_model(), b, **kw)
    for _ in range(3):
        e1.step(b)
    r1 = CloverRunner(e1, work_dir=str(tmp_path), max_epochs=1)
    r1.save_checkpoint(str(tmp_path), 'a.pth')
    nxt = [e1.step(b)['log_vars']['loss'] for _ in range(2)]

    m2 = make_model()
    with torch.no_grad():
        for p in m2.parameters():
            p.mul_(0.5)                                   # start from different weights: the load must matter
    e2 = CloverEngine(m2, b, **kw)
    CloverRunner(e2, work_dir=str(tmp_path)).load_checkpoint(str(tmp_path / 'a.pth'))
    for sg in e2.segments:
        assert torch.equal(sg.shadow, sg.flat_p.to(sg.shadow.dtype))
    l2 = e2.step(b)['log_vars']['loss']
    assert abs(l2 - nxt[0]) < 2e-2 * max(1.0, abs(nxt[0])), (l2, nxt)       # same weights -> same loss

    e3 = CloverEngine(make_model(), b, **kw)
    r3 = CloverRunner(e3, work_dir=str(tmp_path))
    r3.resume(str(tmp_path / 'a.pth'))
    assert e3.step_count == 3 and e3.adam_steps() == 3 and e3.lr_iter == 3
    got = [e3.step(b)['log_vars']['loss'] for _ in range(2)]
    for a, g in zip(nxt, got):
        assert abs(a - g) < 2e-2 * max(1.0, abs(a)), (nxt, got)

    ft = make_finetune_model()
    eft = CloverEngine(ft, b, lr=2e-4, weight_decay=0.0, grad_clip=5.0, max_iters=10 ** 9)
    _, res = CloverRunner(eft, work_dir=str(tmp_path)).load_checkpoint(str(tmp_path / 'a.pth'))
    assert not [k for k in res.missing_keys if 'relative_position_index' not in k]
    assert res.unexpected_keys and all(k.startswith(('mlm_head.', 'mlm_ssl_V_head.', 'mlm_ssl_T_head.'))
                                       for k in res.unexpected_keys)
    p1, pf = dict(e1.model.named_parameters()), dict(ft.named_parameters())
    # e1 has moved on by two steps since the save; the text encoder moved by at most 2 * lr per weight
    k = 'text_backbone.bert.encoder.layer.1.attention.self.query.weight'
    assert (p1[k] - pf[k]).abs().max().item() <= 2.5 * 2e-4


# ----------------------------------------------------------------------------- runner + CLI on the GPU
def test_runner_two_loader_epoch_on_the_engine(tmp_path):
    """CloverRunner.run() (clover_runner.py:76-93 interleave) driving CloverEngine in hipGraph mode over a video
    loader and an image loader, 2 batch indices each: hook order, the reference's log_vars keys, one LR per batch
    index shared by both loaders' steps, four optimizer steps, one graph set per batch geometry, a checkpoint in
    the reference's layout that resumes."""
    from clover_amd.engine import CloverEngine, cosine_lr
    from clover_amd.runner import CheckpointHook, CloverRunner, Hook, LogHook, LrUpdaterHook
    vid = [batch(2, f'run_v{i}') for i in range(2)]
    img = [{k: v.to(DEV) for k, v in cf.cf_batch(2, frames=1, tag=f'run_i{i}').items()} for i in range(2)]
    m = make_model()
    eng = CloverEngine(m, vid[0], lr=1e-3, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
    eng.dry_step(vid[0])
    assert eng.capture(vid[0])
    events = []

    class Rec(Hook):
        def before_run(self, r): events.append('before_run')
        def before_train_epoch(self, r): events.append('before_epoch')
        def before_train_iter(self, r): events.append(('before_iter', r.iter, r.inner_iter))
        def after_train_iter(self, r): events.append(('after_iter', r.iter, sorted(dict(r.outputs['log_vars']))))
        def after_train_epoch(self, r): events.append('after_epoch')
        def after_run(self, r): events.append('after_run')

    r = CloverRunner(eng, model=m, work_dir=str(tmp_path), max_epochs=1)
    lrh = LrUpdaterHook(1e-3, min_lr_ratio=1e-3, warmup='linear', warmup_iters=1, warmup_ratio=0.001, warmup_by_epoch=True)
    log = LogHook(interval=1)
    for h in (lrh, Rec(), log, CheckpointHook(str(tmp_path), 1)):
        r.register_hook(h)
    r.run([vid, img], [('train', 1)], 1)
    torch.cuda.synchronize()
    assert events[0] == 'before_run' and events[1] == 'before_epoch' and events[-2:] == ['after_epoch', 'after_run']
    iters = [e for e in events if isinstance(e, tuple)]
    assert [e[0] for e in iters] == ['before_iter', 'after_iter'] * 4
    assert [e[1] for e in iters if e[0] == 'before_iter'] == [0, 0, 1, 1]          # runner.iter counts batch indices
    keys = ['loss', 'mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'rank_v_vm_loss', 'v_nce_loss']
    assert all(e[2] == keys for e in iters if e[0] == 'after_iter')
    want = [cosine_lr(1e-3, it, 2, 1e-3, 2, 0.001) for it in (0, 0, 1, 1)]          # warm-up: 1 epoch x 2 indices
    assert lrh.history == pytest.approx(want) and eng.last_lr == pytest.approx(want[-1])
    assert eng.adam_steps() == 4 and len(eng._captures) == 2
    assert len(log.records) == 4 and all(torch.isfinite(torch.tensor(rec['loss'])) for rec in log.records)
    ck = torch.load(str(tmp_path / 'epoch_1.pth'), map_location='cpu')
    assert set(ck) == {'meta', 'state_dict', 'optimizer'} and ck['meta']['epoch'] == 1 and ck['meta']['iter'] == 2
    m2 = make_model()
    eng2 = CloverEngine(m2, vid[0], lr=1e-3, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
    r2 = CloverRunner(eng2, model=m2, max_epochs=2)
    r2.resume(str(tmp_path / 'epoch_1.pth'))
    assert r2.epoch == 1 and r2.iter == 2 and eng2.adam_steps() == 4


def test_tools_train_cli_two_loaders(tmp_path):
    """tools/train.py (tools/train.py:26-88,259-340 of the reference) end to end in a child process: config file with
    `_base_`, --cfg-options overrides, linear LR scaling, hipGraph capture per geometry, log lines and the checkpoint."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    syn = "data.synthetic=[{'length':2,'frames':8,'tokens':32},{'length':2,'frames':1,'tokens':32}]"
    cmd = [sys.executable, os.path.join(root, 'tools', 'train.py'), os.path.join(root, 'configs', 'pretrain_synthetic.py'),
           '--launcher', 'none', '--work_dir', str(tmp_path), '--seed', '3',
           '--cfg-options', 'total_epochs=1', 'videos_per_gpu=2', 'log_config.interval=1', syn]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [ln for ln in out.stdout.splitlines() if "'mlm_loss'" in ln]
    assert len(recs) == 4, out.stdout[-2000:]
    assert os.path.exists(os.path.join(str(tmp_path), 'epoch_1.pth'))


def test_first_touch_gradients_match_cleared_slabs(monkeypatch):
    """Weights whose gradient arrives through ONE weight-gradient launch per step are not cleared between steps: that
    launch stores (engine._setup_first_touch).  Against an engine that clears everything (CLOVER_GRAD_FIRST_TOUCH=0) on
    the same weights: the same gradients from an eager backward and from the captured hipGraphs — twice in a row, the
    second time over the first one's stale content — and the same loss trajectory; between steps the uncleared slots
    hold the last gradient (never zero-filled), and a slot a backward does not reach is cleared like the others."""
    from clover_amd.engine import CloverEngine
    b = batch(tag='ft')
    grads, traj = {}, {}
    for ft in ('0', '1'):
        monkeypatch.setenv('CLOVER_GRAD_FIRST_TOUCH', ft)
        for mode in ('eager', 'graph'):
            eng = CloverEngine(make_model(), b, lr=1e-4, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
            if ft == '1':
                assert eng.first_touch_params > 0.5 * eng.num_params, (eng.first_touch_params, eng.num_params)
                assert sum(v.numel() for v in eng._zero_views) + eng.first_touch_params == \
                    sum(sg.flat_g.numel() for sg in eng.segments)
            else:
                assert eng.first_touch_params == 0 and eng._zero_views is None
            if mode == 'graph':
                eng.dry_step(b)
                assert eng.capture(b)
            got = []
            for _ in range(2):
                eng.zero_grads()
                if mode == 'graph':
                    eng._graphed_forward_backward(b)
                else:
                    out = eng.model.train_step(b, None)
                    eng._backward(lambda: out['loss'].backward())
                got.append(torch.cat([sg.flat_g for sg in eng.segments]).clone())
            grads[ft, mode] = got
            eng.zero_grads()
            traj[ft, mode] = [float(eng.step(b)['log_vars']['loss']) for _ in range(4)]
            if ft == '1':
                # between steps the first-touch slots keep the last gradient; everything else is zero
                assert all(float(v.abs().max()) == 0.0 for v in eng._zero_views)
                assert any(float(sg.flat_g.abs().max()) > 0 for sg in eng.segments)
                # a first-touch slot the backward does not write counts as stale (cleared before the optimizer)
                key, (sink, si, off, n) = next(iter(eng._fresh_sinks.items()))
                eng._ft.done.clear()
                assert any(v.data_ptr() == sink.data_ptr() for v in eng._stale_sinks())
                eng._ft.done.add(key)
                assert not any(v.data_ptr() == sink.data_ptr() for v in eng._stale_sinks())
                eng._ft.done.clear()
    for mode in ('eager', 'graph'):
        ref = grads['0', mode][0]
        for g in grads['1', mode] + grads['0', mode][1:]:
            assert bool(torch.isfinite(g).all())
            assert float((g - ref).norm()) <= 1e-3 * float(ref.norm()), (mode, float((g - ref).norm()), float(ref.norm()))
        for x, y in zip(traj['1', mode], traj['0', mode]):
            assert abs(x - y) <= 2e-2 * abs(y), (mode, traj)


def test_first_touch_sink_semantics():
    """ops.linear_wgrad on a first-touch sink: the first call after the state is cleared STORES over stale content, the
    second accumulates; grouped (deferred) and stand-alone launches, in-place and partial + fold shapes, library shapes."""
    from clover_amd import ops
    torch.manual_seed(5)
    for (M, N, K) in ((512, 768, 768), (3136, 768, 3072), (12544, 384, 384), (256, 1000, 768)):
        dy = (torch.randn(M, N, device=DEV) * 0.1).to(HALF)
        x = (torch.randn(M, K, device=DEV) * 0.1).to(HALF)
        ref = dy.float().t() @ x.float()
        for deferred in (False, True):
            sink = torch.full((N, K), float('nan'), device=DEV)          # stale content: must never be read
            bsink = torch.zeros(N, device=DEV)
            st = ops.FirstTouch()
            st.on = True
            sink._clv_ft = st
            for rep in (1, 2):
                if deferred:
                    with ops.defer_folds():
                        ops.linear_wgrad(dy, x, True, sink, bsink)
                else:
                    ops.linear_wgrad(dy, x, True, sink, bsink)
                err = float((sink - rep * ref).abs().max())
                assert err <= 2e-3 * rep * float(ref.abs().max()), (M, N, K, deferred, rep, err)
            assert float((bsink - 2 * dy.float().sum(0)).abs().max()) <= 2e-3 * float(dy.float().sum(0).abs().max()) + 1e-3


class _TwoPathToy(torch.nn.Module):
    """A recognizer-shaped toy whose batch geometry decides which Linear weights the step reaches: `a` always, `b` only
    for the wide geometry (rows == 128) — the situation of ADVICE r3 (first-touch slots across a geometry switch)."""
    CLV_ENCODE_KEYS = ()

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.a = torch.nn.Linear(128, 128)
        self.b = torch.nn.Linear(128, 128)
        with torch.no_grad():
            for lin in (self.a, self.b):
                lin.weight.copy_(torch.randn(128, 128, generator=g) * 0.05)
                lin.bias.zero_()
        self.lazy_log_vars = True

    def encode(self, imgs, video_cut=None, **kw):
        from clover_amd import ops
        x = imgs.to(HALF)
        y = ops.linear(x, self.a.weight, self.a.bias)
        if imgs.shape[0] == 128:
            y = y + ops.linear(x, self.b.weight, self.b.bias)
        return y.float(), None

    def contrastive_losses(self, emb, mlm_loss, gathered=None):
        return dict(toy_loss=(emb * emb).mean())

    def _parse_losses(self, losses, reduce=True):
        from clover_amd.recognizers.base import BaseRecognizer
        return BaseRecognizer._parse_losses(self, losses, reduce)

    def train_step(self, data_batch, optimizer=None, **kw):
        emb, _ = self.encode(data_batch['imgs'])
        loss, log_vars = self._parse_losses(self.contrastive_losses(emb, None))
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data_batch['imgs']))


@pytest.mark.parametrize('mode', ['eager', 'graph', 'eager_dp'])
def test_first_touch_slots_across_a_geometry_switch(mode, monkeypatch):
    """a first-touch slot that geometry A's backward writes and geometry B's never does must not carry A's
    gradient into B's optimizer step.  Alternating A / B steps against an engine that clears every slab
    (CLOVER_GRAD_FIRST_TOUCH=0): identical parameters afterwards, and `b`'s gradient slot is zero in every B step.
    `eager_dp`: the same in eager data-parallel mode on a real 1-rank RCCL group with the bf16 wire — the
    unreached slot must be cleared BEFORE reducer.finish() packs and all-reduces its bucket, because the norm / AdamW
    kernels read the reduced wire copy (engine.finish_backward)."""
    import os
    import torch.distributed as dist
    from clover_amd.engine import CloverEngine
    torch.manual_seed(3)
    wide = dict(imgs=torch.randn(128, 128, device=DEV))
    narrow = dict(imgs=torch.randn(64, 128, device=DEV))
    final = {}
    dp = mode == 'eager_dp'
    if dp:
        monkeypatch.setenv('CLOVER_FORCE_COLLECTIVES', '1')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29563')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        for ft in ('0', '1'):
            monkeypatch.setenv('CLOVER_GRAD_FIRST_TOUCH', ft)
            m = _TwoPathToy().to(DEV)
            eng = CloverEngine(m, wide, lr=1e-2, weight_decay=0.0, grad_clip=0.0, max_iters=10 ** 9)
            if ft == '1':
                assert eng.first_touch_params >= 2 * 128 * 128, eng.first_touch_params
            if dp:
                assert eng.reducer.active and eng.wire is not None
            if mode == 'graph':
                eng.dry_step(wide)
                assert eng.capture(wide)
            for it in range(3):
                eng.step(wide)
                if mode == 'graph':
                    # B's replayed backward must see a clean slot for `b` (checked between the replay and the optimizer)
                    sig = eng._signature(narrow)
                    if sig in eng._captures:
                        eng._activate(sig)
                        eng._graphed_forward_backward(narrow)
                        assert float(m.b.weight.grad.abs().max()) == 0.0, it
                        eng.zero_grads()
                before = m.b.weight.detach().clone()
                eng.step(narrow)
                if dp:
                    # grad_clip = 0, weight_decay = 0: an unreached weight with a zero gradient only moves by Adam's
                    # decaying momentum; with last step's gradient on the wire it would move like a reached one.  The
                    # comparison with the all-cleared engine below is the exact check; this one localises a failure.
                    torch.cuda.synchronize()
                    assert torch.isfinite(m.b.weight).all() and before.shape == m.b.weight.shape
            final[ft] = {n: p.detach().clone() for n, p in m.named_parameters()}
            if mode == 'graph':
                assert len(eng._captures) == 2
            if dp and ft == '1':
                # a caller-driven backward that skips finish_backward() must not silently exchange stale slots
                eng._ft.done.clear()
                eng.model.train_step(narrow, None)['loss'].backward()
                eng.reducer.finish()
                with pytest.raises(RuntimeError, match='finish_backward'):
                    eng.optimizer_step()
    finally:
        if dp:
            dist.destroy_process_group()
    for n in final['0']:
        assert torch.allclose(final['0'][n], final['1'][n], rtol=0, atol=1e-6), n


@pytest.mark.usefixtures('strict_own_gemm')
def test_engine_own_decoder_equals_plain_autograd():
    """BASELINE config 2's model (vocabulary 30522: not a multiple of 8) under the engine: the MLM decoder runs on the own
    GEMM kernels over phantom-padded parameters (engine._Segment, ops.mlm_decoder); losses and the decoder / transform /
    fusion gradients equal those of plain autograd on an identical model (which takes the generic Linear route), and the
    phantom rows stay zero through optimizer steps."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import clover_amd
    from clover_amd.engine import CloverEngine
    torch.manual_seed(21)
    cfg = bench.model_cfg('T', 8)
    m1 = clover_amd.build_model(cfg).to(DEV).eval()
    m2 = clover_amd.build_model(cfg).to(DEV).eval()
    m2.load_state_dict(m1.state_dict())
    b = {k: v.to(DEV) for k, v in bench.synthetic_batch(2, 8, 32, seed=9).items()}
    eng = CloverEngine(m1, b, lr=1e-4, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
    dec = m1.mlm_head.predictions.decoder
    assert hasattr(dec.weight, '_clv_pad_shadow') and dec.weight._clv_pad_shadow.shape[0] == 30528
    o1 = m1.train_step(b, None)
    o1['loss'].backward()
    o2 = m2.train_step(b, None)
    o2['loss'].backward()
    for k in o2['log_vars']:
        assert abs(o1['log_vars'][k] - o2['log_vars'][k]) < 2e-2, k
    p1, p2 = dict(m1.named_parameters()), dict(m2.named_parameters())
    for n in ['mlm_head.predictions.decoder.weight', 'mlm_head.predictions.decoder.bias',
              'mlm_head.predictions.transform.dense.weight', 'multimodal_backbone.bert_encoder.layer.2.output.dense.weight']:
        ref = p2[n].grad.float()
        err = (p1[n].grad.float() / eng.loss_scale - ref).abs().max().item() / (ref.abs().max().item() + 1e-12)
        assert err < 3e-2, (n, err)
    assert float(dec.weight._clv_pad_grad[30522:].abs().max()) == 0.0
    for _ in range(2):
        eng.step(b)
    assert float(dec.weight._clv_pad_weight[30522:].abs().max()) == 0.0 and float(dec.bias._clv_pad_weight[30522:].abs().max()) == 0.0
    assert float(dec.weight._clv_pad_shadow_t[:, 30522:].abs().max()) == 0.0
