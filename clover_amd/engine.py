"""Data-parallel training engine for the Clover pre-training step on MI355X.

What the reference spreads over MMDistributedDataParallel (tools/train.py:146-154), mmcv's
DefaultOptimizerConstructor + AdamW (pretrain_webvid_cc3m.py:129-137), Fp16OptimizerHook
(mmcv_Fp16OptimizerHook.py:96-149) and the CosineAnnealing LR hook (:139-140) is one object:

* one process per GPU; parameters, gradients and Adam moments live in FLAT fp32 buffers
  (decay / no-decay segments), so the gradient exchange is a handful of large RCCL
  all-reduces over xGMI instead of per-tensor traffic, launched from autograd hooks as soon
  as a bucket is complete (overlapping the rest of backward), and exactly once per step —
  the reference all-reduces twice (DDP reducer + allreduce_grads);
* gradient averaging (the 1/W that gives the reference's R6 behaviour), global-norm clipping
  (max_norm 15) and AdamW run as two HIP kernels per segment (clv_sumsq, clv_adamw_step) with
  no host synchronisation — the reference syncs the host once per parameter for its overflow
  check (fp16_utils.py:328-349); bf16 needs no loss scaling, non-finite norms skip the step;
* statically unused parameters (BERT pooler, the fusion model's own embeddings — the reason
  the reference needs find_unused_parameters=True) are detected on a dry run and excluded.
"""
import math

import os

import torch
import torch.distributed as dist

from .utils.dist import collectives_active
import torch.nn as nn

from . import ops
from .utils.grad_reducer import BucketedGradReducer


def paramwise_weight_decay(model, base_wd, norm_decay_mult=0.0, bias_decay_mult=0.0, custom_keys=None):
    """mmcv DefaultOptimizerConstructor rules: custom_keys by substring
    (longest first) > norm layers > bias > default.  Returns {param_name: weight_decay}."""
    custom_keys = custom_keys or {}
    sorted_keys = sorted(sorted(custom_keys.keys()), key=len, reverse=True)
    norm_types = (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm, nn.GroupNorm, nn.LayerNorm)
    out = {}
    for mod_name, mod in model.named_modules():
        for pname, p in mod.named_parameters(recurse=False):
            full = f'{mod_name}.{pname}' if mod_name else pname
            wd = base_wd
            hit = False
            for key in sorted_keys:
                if key in full:
                    wd = base_wd * custom_keys[key].get('decay_mult', 1.0)
                    hit = True
                    break
            if not hit:
                if isinstance(mod, norm_types):
                    wd = base_wd * norm_decay_mult
                elif pname == 'bias':
                    wd = base_wd * bias_decay_mult
            out[full] = wd
    return out


def paramwise_options(model, base_wd, paramwise_cfg=None):
    """mmcv 1.3.x DefaultOptimizerConstructor.add_params, per parameter:
    {name: (weight_decay, lr_mult)}.  A ``custom_keys`` hit (substring, longest key first) decides both
    ``lr_mult`` and ``decay_mult`` and switches the other rules off; otherwise ``bias_lr_mult`` applies to every
    ``bias`` outside norm layers, and the decay is ``norm_decay_mult`` for norm layers, else ``bias_decay_mult`` for
    a ``bias``, else 1."""
    cfg = paramwise_cfg or {}
    custom_keys = cfg.get('custom_keys') or {}
    sorted_keys = sorted(sorted(custom_keys.keys()), key=len, reverse=True)
    norm_types = (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm, nn.GroupNorm, nn.LayerNorm)
    out = {}
    for mod_name, mod in model.named_modules():
        is_norm = isinstance(mod, norm_types)
        for pname, p in mod.named_parameters(recurse=False):
            full = f'{mod_name}.{pname}' if mod_name else pname
            wd, lr_mult = base_wd, 1.0
            for key in sorted_keys:
                if key in full:
                    lr_mult = custom_keys[key].get('lr_mult', 1.0)
                    wd = base_wd * custom_keys[key].get('decay_mult', 1.0)
                    break
            else:
                if pname == 'bias' and not is_norm:
                    lr_mult = cfg.get('bias_lr_mult', 1.0)
                if is_norm:
                    wd = base_wd * cfg.get('norm_decay_mult', 1.0)
                elif pname == 'bias':
                    wd = base_wd * cfg.get('bias_decay_mult', 1.0)
            out[full] = (float(wd), float(lr_mult))
    return out


def cosine_lr(base_lr, it, max_iters, min_lr_ratio=1e-3, warmup_iters=0, warmup_ratio=1e-3):
    """mmcv CosineAnnealing (by_epoch=False) with linear warm-up."""
    end = base_lr * min_lr_ratio
    lr = end + 0.5 * (base_lr - end) * (1 + math.cos(math.pi * min(it, max_iters) / max(1, max_iters)))
    if it < warmup_iters:
        k = (1 - it / warmup_iters) * (1 - warmup_ratio)
        lr = lr * (1 - k)
    return lr


class _Segment:
    """A flat fp32 slab: params / grads / exp_avg / exp_avg_sq of one (weight_decay, lr_mult) class — what mmcv
    expresses as one param group per parameter collapses to one slab per DISTINCT option pair."""

    def __init__(self, named_params, weight_decay, device, lr_mult=1.0):
        self.weight_decay = weight_decay
        self.lr_mult = lr_mult
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        # 16-byte aligned slots (fp32 AND bf16 views: the GEMM kernels take 16-byte aligned operands).  A parameter may ask for PHANTOM rows behind its last one (``_clv_pad_rows``: the MLM
        # decoder's [30522, H] weight and [30522] bias -> 30528, a multiple of 64): the padded [rows + pad, ...] views of the
        # fp32 data, the gradient and the bf16 shadow are what its GEMMs run on (16-byte row groups, no edge code); the
        # phantom rows start at zero, receive zero gradients (the loss kernel zeroes the padding columns of d logits) and
        # so stay zero under AdamW; the module-visible parameter is the unpadded view.
        def slot_elems(p):
            pad = int(getattr(p, '_clv_pad_rows', 0) or 0)
            return p.numel() + pad * (p.numel() // p.shape[0] if p.dim() > 0 and p.shape[0] > 0 else 0)
        sizes = [(slot_elems(p) + 7) // 8 * 8 for p in self.params]          # 8 elements: the bf16 shadow slots are 16-byte aligned too
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + s)
        n = self.offsets[-1]
        self.flat_p = torch.zeros(n, device=device, dtype=torch.float32)
        self.flat_g = torch.zeros(n, device=device, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=device, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=device, dtype=torch.float32)
        self.shadow = torch.zeros(n, device=device, dtype=ops.BF16)      # 16-bit compute copy (bf16, or fp16 with CLOVER_HALF=f16)
        for p, off in zip(self.params, self.offsets):
            view = self.flat_p[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            hook = getattr(p, '_clv_unscale', None)        # (the recognizer's loss-scale hook of a plain-autograd parameter:
            if hook is not None:                           # under the engine every gradient in the slab stays scaled)
                hook.remove()
                p._clv_unscale = None
            p.grad = self.flat_g[off:off + p.numel()].view_as(p)
            p._clv_grad = p.grad                                                   # sink for ops.linear
            p._clv_shadow = self.shadow[off:off + p.numel()].view_as(p)
            pad = int(getattr(p, '_clv_pad_rows', 0) or 0)
            if pad:
                shp = (p.shape[0] + pad,) + tuple(p.shape[1:])
                n = slot_elems(p)
                p._clv_pad_weight = self.flat_p[off:off + n].view(shp)
                p._clv_pad_grad = self.flat_g[off:off + n].view(shp)
                p._clv_pad_shadow = self.shadow[off:off + n].view(shp)
        self.shadow.copy_(self.flat_p)
        self.shadow_t, self._t_table = None, None
        self._fused = []                     # fused views handed out (fused_view), for build_transposed()

    def build_transposed(self):
        """bf16 W^T copies for the Linear weights whose input-gradient GEMM is a K-contiguous HIP kernel
        (ops.linear_dgrad): one batched transpose per optimizer step refreshes them all.  A fused view (BERT Q|K|V,
        applied as ONE [3H, H] GEMM) gets the transpose of the fused matrix, [H, 3H]; its members then need none of
        their own.  Called once, after the engine has made the fused views."""
        fused_members = {id(m) for f in self._fused if getattr(f, '_clv_want_t', False) for m in f._clv_members}
        def tshape(p):                       # a phantom-padded weight is transposed with its phantom rows
            pad = int(getattr(p, '_clv_pad_rows', 0) or 0)
            return (p.shape[0] + pad, p.shape[1])
        want = [(p, off, tshape(p)) for p, off in zip(self.params, self.offsets)
                if getattr(p, '_clv_want_t', False) and p.dim() == 2 and id(p) not in fused_members]
        want += [(f, f._clv_off, tuple(f.shape)) for f in self._fused if getattr(f, '_clv_want_t', False) and f.dim() == 2]
        if not want or not self.flat_p.is_cuda:
            return
        device = self.flat_p.device
        self.shadow_t = torch.zeros(sum((sh[0] * sh[1] + 7) // 8 * 8 for _, _, sh in want), device=device, dtype=ops.BF16)
        entries, toff = [], 0
        for p, off, shape in want:
            n = shape[0] * shape[1]
            entries.append((off, toff, shape[0], shape[1]))
            view = self.shadow_t[toff:toff + n].view(shape[1], shape[0])
            if shape[0] != p.shape[0]:
                p._clv_pad_shadow_t = view         # [cols, rows + pad]: only the padded kernels may take it
            else:
                p._clv_shadow_t = view
            toff += (n + 7) // 8 * 8
        self._t_table = ops.transpose_table(entries, device)
        self.refresh_transposed()

    def refresh_transposed(self):
        if self._t_table is not None:
            tab, n, tiles = self._t_table
            ops.transpose_batch(self.shadow, self.shadow_t, tab, n, tiles)

    def fused_view(self, members):
        """One Parameter-like view over `members` (adjacent slots of this slab, equal trailing dims,
        concatenated along dim 0): fp32 data, bf16 shadow and gradient sink are slices of the slabs, so a
        module that applies the members as ONE GEMM (BERT Q|K|V) needs no cat / cast / gradient split."""
        idx = [next(i for i, q in enumerate(self.params) if q is m) for m in members]
        assert idx == list(range(idx[0], idx[0] + len(idx))), 'fusion group is not adjacent in the slab'
        assert all(m.numel() % 8 == 0 and m.shape[1:] == members[0].shape[1:] for m in members)      # slots are 8-element aligned
        off, n = self.offsets[idx[0]], sum(m.numel() for m in members)
        shape = (sum(m.shape[0] for m in members),) + tuple(members[0].shape[1:])
        f = torch.nn.Parameter(self.flat_p[off:off + n].view(shape))
        f._clv_grad = self.flat_g[off:off + n].view(shape)
        f._clv_shadow = self.shadow[off:off + n].view(shape)
        f._clv_members = list(members)
        f._clv_ready = lambda ms=f._clv_members: [m._clv_ready() for m in ms] and None
        f._clv_off = off
        f._clv_want_t = all(getattr(m, '_clv_want_t', False) for m in members)
        self._fused.append(f)
        return f


class CloverEngine:
    def __init__(self, model, sample_batch, lr=5e-5, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.005,
                 paramwise_cfg=None, grad_clip=15.0, max_iters=100000, warmup_iters=0, min_lr_ratio=1e-3,
                 warmup_ratio=1e-3, bucket_mb=64, loss_scale=None):
        self.model = model
        # Loss scaling (mmcv_Fp16OptimizerHook.py:33-50,96-149; `fp16 = dict(loss_scale='dynamic')` in the headline config,
        # pretrain_webvid_cc3m.py:21).  `loss_scale` takes what the hook takes: a float (static), 'dynamic'
        # (LossScaler(mode='dynamic'): 2**32, factor 2, window 1000) or a dict of LossScaler arguments; None = the build's
        # default (_lib.LOSS_SCALE / CLOVER_LOSS_SCALE: static 1024 for fp16, none for bf16).  The scaler lives on the DEVICE
        # next to the optimizer's scalars (ops.optim_state_set_scaler): the recognizer multiplies the root gradient of every
        # backward by it (BaseRecognizer._parse_losses reads model._clv_loss_scale_dev), the gradients in the slabs are
        # scaled, clv_optim_prep divides the scale out of the norm and the update, skips the step on an overflow and moves a
        # dynamic scale — no host round trip, and a captured step follows the scale without re-capture.
        from . import _lib as _clv_lib
        if loss_scale is None:
            loss_scale = _clv_lib.LOSS_SCALE_SPEC
        if loss_scale == 'dynamic':
            self.scaler_cfg = dict(init_scale=2.0 ** 32, mode='dynamic', scale_factor=2.0, scale_window=1000)
        elif isinstance(loss_scale, dict):
            self.scaler_cfg = dict(dict(init_scale=2.0 ** 32, mode='dynamic', scale_factor=2.0, scale_window=1000), **loss_scale)
        elif isinstance(loss_scale, (int, float)):
            self.scaler_cfg = dict(init_scale=float(loss_scale), mode='static', scale_factor=2.0, scale_window=1000)
        else:
            raise ValueError(f'loss_scale must be of type float, dict, or "dynamic", got {loss_scale}')
        assert self.scaler_cfg['mode'] in ('dynamic', 'static'), 'mode can only be dynamic or static'
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.base_lr, self.betas, self.eps = lr, betas, eps
        self.grad_clip = grad_clip
        self.max_iters, self.warmup_iters = max_iters, warmup_iters
        self.min_lr_ratio, self.warmup_ratio = min_lr_ratio, warmup_ratio
        self.step_count = 0                    # optimizer_step() calls (taken or skipped)
        self._stale_cleared = False            # finish_backward() cleared the unreached first-touch slots of this step
        self.lr_iter = 0                       # index into the LR schedule when no runner drives it (set_lr)
        self._lr_external = None
        self.graph = None
        self.graph_bwd_video = None
        self.graph_bwd_text = None
        self._captures = {}                    # batch-shape signature -> the captured graphs + their static tensors
        self._active_sig = None
        device = next(model.parameters()).device
        if device.type != 'cuda':
            raise RuntimeError('CloverEngine drives the HIP kernels: move the model to an MI355X first (there is no '
                               f'CPU path; got parameters on {device})')
        pw = paramwise_cfg or dict(norm_decay_mult=0.0, bias_decay_mult=0.0,
                                   custom_keys={'absolute_pos_embed': dict(decay_mult=0.),
                                                'relative_position_bias_table': dict(decay_mult=0.)})
        opt_map = paramwise_options(model, weight_decay, pw)       # name -> (weight_decay, lr_mult)

        # ---- dry run: which parameters does the step graph actually reach?
        model.zero_grad(set_to_none=True)
        out = model.train_step(sample_batch, None)
        out['loss'].backward()
        used = [(n, p) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
        name_of = {id(p): n for n, p in model.named_parameters()}
        self.unused_names = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
        model.zero_grad(set_to_none=True)

        # reverse registration order ~ order in which backward produces the gradients
        used = list(reversed(used))
        # modules that apply several parameters as one GEMM ask for adjacent slab slots (clv_fuse_groups())
        groups = []
        used_ids = {id(p) for _, p in used}
        for mod in model.modules():
            for grp in (mod.clv_fuse_groups() if hasattr(mod, 'clv_fuse_groups') else []):
                if all(id(q) in used_ids for q in grp) and len({opt_map[name_of[id(q)]] for q in grp}) == 1:
                    groups.append((mod, grp))
        member = {id(q): gi for gi, (_, grp) in enumerate(groups) for q in grp}
        ordered, placed = [], set()
        for n, p_ in used:
            gi = member.get(id(p_))
            if gi is None:
                ordered.append((n, p_))
            elif gi not in placed:
                placed.add(gi)
                ordered.extend((name_of[id(q)], q) for q in groups[gi][1])
        used = ordered
        # one slab per distinct (weight_decay, lr_mult): the reference config has two (decayed weights; norm / bias /
        # position-table parameters without decay), any other decay_mult / lr_mult of a paramwise_cfg adds its own
        classes = {}
        for n, p_ in used:
            classes.setdefault(opt_map[n], []).append((n, p_))
        self.segments = [_Segment(members, wd, device, lr_mult=lm)
                         for (wd, lm), members in sorted(classes.items(), key=lambda kv: (-kv[0][0], kv[0][1]))]
        # gradient-norm accumulator: [0] the sum of squares the norm kernels add to; from float 16 on, the CLV_SUMSQ_SLOTS
        # partial sums (64 bytes apart) that the weight-gradient kernels fill in fused-norm mode (_setup_fused_norm)
        self.sumsq = torch.zeros(16 + 16 * ops.SUMSQ_SLOTS, device=device, dtype=torch.float32)
        self._norm_tables = None               # per segment (table, blocks) of what the norm pass still reads; None: whole slabs
        self.optim_state = ops.optim_state_new(device)
        self._scaler_on = self.scaler_cfg['mode'] == 'dynamic' or self.scaler_cfg['init_scale'] != 1.0
        if self._scaler_on:
            ops.optim_state_set_scaler(self.optim_state, self.scaler_cfg['init_scale'], self.scaler_cfg['mode'] == 'dynamic',
                                       self.scaler_cfg['scale_factor'], self.scaler_cfg['scale_window'])
        object.__setattr__(model, '_clv_loss_scale_dev', self.optim_state.view(torch.float32)[8] if self._scaler_on else False)     # False: an engine without a scaler
        self.num_params = sum(p.numel() for _, p in used)

        # ---- gradient buckets (contiguous slices of the flat grad buffers) + readiness hooks
        # gradient classes in the order their backward completes: 'h' heads + fusion encoder, 't' text encoder,
        # 'v1' late video stages, 'v0' early video stages + patch embedding (graph mode sends buckets per class)
        self._pclass = {}
        if hasattr(model, 'backbone'):
            for q in model.backbone.parameters():
                self._pclass[id(q)] = 'v0'
            if hasattr(model.backbone, 'late_parameters'):
                for q in model.backbone.late_parameters():
                    self._pclass[id(q)] = 'v1'
        if hasattr(model, 'text_backbone'):
            for q in model.text_backbone.parameters():
                self._pclass[id(q)] = 't'
        # data-parallel jobs exchange the gradients as bf16 (CLOVER_BF16_ALLREDUCE=0: fp32): each bucket's slice is packed
        # into a bf16 wire slab right before its all-reduce and sumsq / AdamW read the reduced bf16 copy — 427 MB instead
        # of 757 MB per step over xGMI at config 2; accumulation in the backward and the update itself stay fp32
        self.wire = None
        if collectives_active() and os.environ.get('CLOVER_BF16_ALLREDUCE', '1') == '1':
            self.wire = [torch.zeros_like(seg.flat_g, dtype=ops.BF16) for seg in self.segments]
        self.reducer = BucketedGradReducer([(seg.flat_g, seg.params, seg.offsets) for seg in self.segments],
                                           bucket_bytes=bucket_mb << 20,
                                           split_key=lambda q: self._pclass.get(id(q), 'h'), wire=self.wire)
        per_mod = {}
        for mod, grp in groups:
            seg = next(sg for sg in self.segments if any(q is grp[0] for q in sg.params))
            per_mod.setdefault(id(mod), (mod, []))[1].append(seg.fused_view(grp))
        for mod, views in per_mod.values():
            mod._clv_fused = views                       # same order as clv_fuse_groups()
        for seg in self.segments:
            seg.build_transposed()
        self._zero_views = None                # None: clear whole slabs
        self._stale_views = []
        self._prepacked = frozenset()
        self._setup_first_touch(sample_batch)
        self._setup_fused_norm()

    def _setup_fused_norm(self):
        """One-rank jobs: the first-touch weight gradients (most of the parameters: each is written by exactly one launch per
        step) deliver their sum of squares from the kernel that writes them (ops.linear_wgrad: flag bit 2 of the grouped /
        fold launches, an explicit pass on any other path), so the norm pass before the clip reads only the REST of the
        slabs — `clv_sumsq_ranges` over the complement — instead of all 0.76 GB.  Not in data-parallel jobs: their norm is
        that of the REDUCED gradients (mmcv_Fp16OptimizerHook.py:119-131).  CLOVER_FUSED_NORM=0 switches it off."""
        fresh = getattr(self, '_fresh_sinks', None)
        if (not fresh or self.reducer.active or self.wire is not None or os.environ.get('CLOVER_FUSED_NORM', '1') != '1'
                or not self.segments[0].flat_g.is_cuda):
            return
        slots = self.sumsq[16:]
        tables = []
        for si, seg in enumerate(self.segments):
            spans = sorted((off, off + n) for _, sj, off, n in fresh.values() if sj == si)
            ranges, pos = [], 0
            for a, b in spans:
                if a > pos:
                    ranges.append((pos, a))
                pos = max(pos, b)
            if pos < seg.flat_g.numel():
                ranges.append((pos, seg.flat_g.numel()))
            tables.append(ops.sumsq_range_table(ranges, seg.flat_g.device))
        self._norm_state = ops.FusedNormState()
        for sk, _, _, _ in fresh.values():
            sk._clv_ssq = (self.sumsq, slots, self._norm_state)
        self._norm_tables = tables

    # ------------------------------------------------------------------ gradient clearing
    def _setup_first_touch(self, sample_batch):
        """Decide which weight gradients are never cleared.  A Linear weight whose gradient reaches its slab slot through
        exactly ONE weight-gradient launch per step (ops.linear_wgrad) needs neither the zero-fill after the optimizer
        nor the read of a "+=": that launch STORES (ops.first_touch) — 8 B of HBM traffic per parameter and step less on
        most of the 189 M parameters.  Which slots qualify is MEASURED, not declared: a reference backward into cleared
        slabs, then the same backward (same RNG state) into slabs poisoned with NaN with every once-written sink marked
        first-touch; a slot qualifies when it comes out finite and equal to the reference (a gradient that also arrives
        through autograd BEFORE its launch, or a sink the launch does not recognise, fails here and keeps being cleared).
        A third pass with the final marking must reproduce every gradient, otherwise the scheme is switched off."""
        self._ft = ops.FirstTouch()
        self.first_touch_params = 0
        if os.environ.get('CLOVER_GRAD_FIRST_TOUCH', '1') != '1':
            return
        sinks = {}                             # data_ptr -> (sink tensor, segment index, slab offset, numel)
        for si, seg in enumerate(self.segments):
            for q, off in zip(seg.params, seg.offsets):
                sk = getattr(q, '_clv_pad_grad', None)         # a phantom-padded parameter's kernels write the padded view
                sk = sk if sk is not None else q._clv_grad
                sinks[sk.data_ptr()] = (sk, si, off, sk.numel())
            for f in seg._fused:               # a fused view starts where its first member does: the view is what ops sees
                sinks[f._clv_grad.data_ptr()] = (f._clv_grad, si, f._clv_off, f.numel())
        cpu_rng, gpu_rng = torch.get_rng_state(), torch.cuda.get_rng_state()
        drop_ctr = ops._dropout_counter(self.segments[0].flat_g.device)      # the kernels' own dropout seeds
        drop_ctr0 = drop_ctr.clone()

        def mark(ptrs):
            for t, _, _, _ in sinks.values():
                t._clv_ft = None
            for ptr in ptrs:
                sinks[ptr][0]._clv_ft = self._ft
            self._ft.on = bool(ptrs)
            self._ft.done.clear()

        def span(slabs, ptr):
            _, si, off, n = sinks[ptr]
            return slabs[si][off:off + n]

        def run(ptrs):
            torch.set_rng_state(cpu_rng)
            torch.cuda.set_rng_state(gpu_rng)
            drop_ctr.copy_(drop_ctr0)
            for seg in self.segments:
                seg.flat_g.zero_()
            mark(ptrs)
            for ptr in ptrs:
                span([seg.flat_g for seg in self.segments], ptr).fill_(float('nan'))
            out = self.model.train_step(sample_batch, None)
            self._backward(lambda: out['loss'].backward())
            return [seg.flat_g.clone() for seg in self.segments]

        def same(a, b):
            return bool(torch.isfinite(a).all()) and float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-12

        hooks_were = self.reducer.enabled
        self.reducer.enabled = False
        ok = []
        try:
            ops.FRESH_LOG = {}
            ref = run([])
            log, ops.FRESH_LOG = ops.FRESH_LOG, None
            cand = [ptr for ptr, (calls, numel) in log.items() if calls == 1 and ptr in sinks and sinks[ptr][3] == numel]
            dbg = os.environ.get('CLOVER_FT_DEBUG') == '1'
            if dbg:
                print(f'[first-touch] sinks {len(sinks)}  logged {len(log)}  candidates {len(cand)}', flush=True)
            # exactly three passes on every rank, whatever they find: the step contains collectives (feature all-gather)
            got = run(cand)
            ok = [ptr for ptr in cand if same(span(got, ptr), span(ref, ptr))]
            if dbg:
                bad = [ptr for ptr in cand if ptr not in ok]
                print(f'[first-touch] pass {len(ok)}  fail {len(bad)}', flush=True)
                for ptr in bad[:6]:
                    a, r = span(got, ptr), span(ref, ptr)
                    print('   fail numel', sinks[ptr][3], 'finite', bool(torch.isfinite(a).all()),
                          'relerr', float((a - r).norm() / (r.norm() + 1e-30)), flush=True)
            got = run(ok)
            if dbg:
                for si, (g, r) in enumerate(zip(got, ref)):
                    print(f'[first-touch] validation slab {si}: finite {bool(torch.isfinite(g).all())} relerr '
                          f'{float((g - r).norm() / (r.norm() + 1e-30)):.3e}', flush=True)
            if not all(same(g, r) for g, r in zip(got, ref)):
                ok = []
        finally:
            ops.FRESH_LOG = None
            self.reducer.reset()
            self.reducer.enabled = hooks_were
            torch.set_rng_state(cpu_rng)
            torch.cuda.set_rng_state(gpu_rng)
            drop_ctr.copy_(drop_ctr0)
        mark(ok)
        for seg in self.segments:
            seg.flat_g.zero_()
        if not ok:
            return
        # what is still cleared every step: the complement of the first-touch slots, as few contiguous views as possible
        self._fresh_sinks = {id(sinks[ptr][0]): sinks[ptr] for ptr in ok}
        self._zero_views = []
        for si, seg in enumerate(self.segments):
            spans = sorted((off, off + n) for _, sj, off, n in self._fresh_sinks.values() if sj == si)
            pos = 0
            for a, b in spans:
                if a > pos:
                    self._zero_views.append(seg.flat_g[pos:a])
                pos = max(pos, b)
            if pos < seg.flat_g.numel():
                self._zero_views.append(seg.flat_g[pos:])
        self.first_touch_params = sum(n for _, _, _, n in self._fresh_sinks.values())

    def zero_grads(self):
        """Clear the gradients for the next backward: zero-fill what accumulates, mark the first-touch slots stale."""
        self._ft.done.clear()
        if self._zero_views is None:
            for seg in self.segments:
                seg.flat_g.zero_()
            return
        views = self._zero_views + self._stale_views
        if self._norm_tables is not None:
            views = views + [self.sumsq]       # a backward without an optimizer step (dry run, warm-up) left partial sums
        if views:
            torch._foreach_zero_(views)

    def _stale_sinks(self):
        """First-touch slots the last backward did NOT write (a weight the batch did not reach): their content is the
        previous step's gradient — they must be cleared like the others."""
        if self._zero_views is None:
            return []
        return [self.segments[si].flat_g[off:off + n]
                for key, (_, si, off, n) in self._fresh_sinks.items() if key not in self._ft.done]

    # ------------------------------------------------------------------ one step
    def current_lr(self):
        """LR of the NEXT optimizer step.  Under a runner the schedule is indexed by the runner's ``iter`` — batch
        indices, which the two-loader mode advances once per TWO optimizer steps (clover_runner.py:76-91) — and set
        through ``set_lr`` by ``runner.LrUpdaterHook`` as mmcv's LR hook does in ``before_train_iter``.  Stand-alone
        (bench.py) the engine indexes the same schedule by its own count of steps, starting at 0 like mmcv's iter."""
        if self._lr_external is not None:
            return self._lr_external
        return cosine_lr(self.base_lr, self.lr_iter, self.max_iters, self.min_lr_ratio, self.warmup_iters,
                         self.warmup_ratio)

    def set_lr(self, lr):
        """Externally scheduled learning rate, used by every following step until changed (None: own schedule)."""
        self._lr_external = None if lr is None else float(lr)

    def dry_step(self, batch):
        """forward + backward + gradient exchange WITHOUT the optimizer: warms the kernels, the GEMM tuner and the
        gradient reducer's hook calibration and leaves weights, Adam moments, step counts and the LR index untouched
        (gradients are zeroed again)."""
        self._ft.done.clear()
        out = self.model.train_step(batch, None)
        self._backward(lambda: out['loss'].backward())
        self.finish_backward()
        self._stale_cleared = False
        for seg in self.segments:
            seg.flat_g.zero_()
        self.sumsq.zero_()
        if self._norm_tables is not None:
            self._norm_state.dirty = False
        self._ft.done.clear()
        return out

    def step(self, batch):
        """forward + backward + gradient all-reduce + clip + AdamW.  Returns train_step's dict."""
        self.reducer.begin_step()
        if self.graph is not None:
            sig = self._signature(batch)
            if sig != self._active_sig:
                # a different batch geometry (the reference alternates 8-frame video and 1-frame image batches,
                # clover_runner.py:76-93): one set of graphs per geometry, captured on first sight
                if sig in self._captures:
                    self._activate(sig)
                else:
                    self.capture(batch)
            out = self._graphed_forward_backward(batch)
        else:
            self.reducer.prepacked = frozenset()
            self._ft.done.clear()
            out = self.model.train_step(batch, None)
            self._backward(lambda: out['loss'].backward())
        self.finish_backward()
        self.optimizer_step()
        self._mark('optimizer')
        return out

    def finish_backward(self):
        """Close a backward pass: clear the first-touch slots an EAGER backward did not reach, THEN complete the gradient
        exchange.  The order matters in data-parallel mode: ``reducer.finish()`` packs and all-reduces the
        bucket of an unreached parameter, and the norm / AdamW kernels read the reduced wire copy — a slot cleared after
        the exchange would still update the weight with the previous step's gradient.  (A captured geometry's unreached
        slots are cleared with the rest of the gradients after every optimizer step: ``_stale_views``.)  A caller that
        drives its own ``loss.backward()`` calls this instead of ``reducer.finish()``."""
        if self.graph is None:
            stale = self._stale_sinks()
            if stale:
                torch._foreach_zero_(stale)
            self._stale_cleared = True
        self.reducer.finish()

    def _backward(self, run):
        """Run a backward pass / segment with the weight-gradient folds deferred to ONE batched launch at its end —
        unless gradient-ready hooks put buckets on the wire from inside the pass (eager data-parallel mode), where a
        gradient must be final when its hook fires."""
        if (self.reducer.active and self.reducer.enabled) or os.environ.get('CLOVER_DEFER_FOLDS', '1') != '1':
            run()
        else:
            with ops.defer_folds():
                run()

    # ------------------------------------------------------------------ hipGraph mode
    _one = None

    def _parse_root_scaled(self, losses, **kw):
        """model._parse_losses for a backward the engine starts itself: the loss scale is handed to autograd as the ROOT
        gradient (_root_gradient) instead of through the recognizer's scaling node — one dependent launch less on the
        step's serial chain (same-box 10.29-10.31 -> see DESIGN §5)."""
        object.__setattr__(self.model, '_clv_root_scaled', True)
        try:
            return self.model._parse_losses(losses, **kw)
        finally:
            object.__setattr__(self.model, '_clv_root_scaled', False)

    def _root_gradient(self, loss):
        """d loss / d loss of an engine-started backward: the device-resident loss scale (a 0-dim view of the optimizer
        record: read when the kernels run, so a captured section follows a moving scale), or a cached 1."""
        if self._scaler_on:
            return self.model._clv_loss_scale_dev
        if self._one is None or self._one.device != loss.device or self._one.dtype != loss.dtype:
            self._one = torch.ones_like(loss)
        return self._one

    _CAPTURE_FIELDS = ('graph', 'graph_bwd', 'graph_bwd_video', 'graph_bwd_text', '_static_batch', '_static_emb',
                       '_static_mlm', '_static_demb', '_static_dmlm', '_static_cuts', 'graph_loss', '_loss_io',
                       '_stale_views', '_prepacked')

    @staticmethod
    def _signature(batch):
        return tuple((k, tuple(v.shape)) for k, v in sorted(batch.items()))

    def _activate(self, sig):
        for f, v in zip(self._CAPTURE_FIELDS, self._captures[sig]):
            setattr(self, f, v)
        self._active_sig = sig
        # The previous optimizer step cleared the stale views of the geometry that ran THEN.  A first-touch slot that
        # geometry writes and this one never does still holds that step's gradient: clear this geometry's stale slots
        # before its first replay (afterwards zero_grads() keeps them clear while it stays active).
        if self._stale_views:
            torch._foreach_zero_(self._stale_views)

    def input_buffers(self):
        """The static input tensors of the active hipGraphs (None before capture): a loader that writes its host-to-device
        copies straight into them — and hands the same dict to step() — saves the per-step staging copy."""
        return self._static_batch if self.graph is not None else None

    def _graphed_forward_backward(self, batch):
        """encode (hipGraph) -> gather + contrastive losses (eager: it holds the step's only forward
        collective) -> backward of the losses (eager, a handful of kernels) -> encode backward (hipGraph)."""
        for k, v in self._static_batch.items():
            if batch[k] is not v:
                v.copy_(batch[k], non_blocking=True)
        self._mark('start')
        self.graph.replay()
        self._mark('forward')
        if getattr(self, 'graph_loss', None) is not None:
            # the loss section as a graph of its own: only the feature all-gather (and the all-reduce of the logged
            # scalars) stay eager — ~60 launch-bound kernels otherwise paced by the host
            g_static, names, packed, loss = self._loss_io
            with torch.no_grad():
                if collectives_active():
                    from .utils.gather_loss import gather_rows
                    g_static.copy_(gather_rows(self._static_emb.float(), equal_sizes=True))
                else:
                    g_static.copy_(self._static_emb)
            self.graph_loss.replay()
            if collectives_active():
                packed = packed / self.world
                dist.all_reduce(packed)
            else:
                packed = packed.clone()        # the graph's static output: the next replay overwrites it
            from .recognizers.base import LazyLogVars
            log_vars = LazyLogVars(names, packed)
            loss = loss.detach().clone()
        else:
            emb = self._static_emb.detach().requires_grad_()
            mlm = self._static_mlm.detach().requires_grad_() if self._static_mlm is not None else None
            losses = self.model.contrastive_losses(emb, mlm)
            loss, log_vars = self._parse_root_scaled(losses)
            loss.backward(gradient=self._root_gradient(loss))
            self._static_demb.copy_(emb.grad)
            if mlm is not None:
                self._static_dmlm.copy_(mlm.grad)
        self._mark('loss')
        self._replay_backward()
        self._mark('backward')
        return dict(loss=loss.detach(), log_vars=log_vars, num_samples=len(next(iter(batch.values()))))

    # Phase clock of the graphed step (bench.py --phases): events on the main stream at the phase boundaries — the
    # forward graph, the eager loss section, the backward graphs, reducer.finish() + optimizer.  GPU time between
    # consecutive marks, averaged by phase_ms(); off (None) unless start_phase_timing() was called.
    _phase_ev = None

    def start_phase_timing(self):
        self._phase_ev = []

    def _mark(self, name):
        if self._phase_ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._phase_ev.append((name, e))

    def phase_ms(self):
        """Mean GPU milliseconds per phase since start_phase_timing() (call after a device sync); stops the clock."""
        ev, self._phase_ev = self._phase_ev, None
        tot, cnt = {}, {}
        for (_, a), (n, b) in zip(ev, ev[1:]):
            if n != 'start':
                tot[n] = tot.get(n, 0.0) + a.elapsed_time(b)
                cnt[n] = cnt.get(n, 0) + 1
        return {n: tot[n] / cnt[n] for n in tot}

    def _ready(self, *classes):
        cl = self._pclass
        return lambda q: cl.get(id(q), 'h') in classes

    def _replay_backward(self):
        self.reducer.prepacked = getattr(self, '_prepacked', None) or frozenset()
        self.graph_bwd.replay()
        if self.graph_bwd_video is None:
            return
        # Data-parallel mode: the backward is cut at the encoders' outputs (and once inside the video encoder).  The
        # heads / fusion gradients are complete now: put their buckets on the wire.  Then the text encoder's
        # backward (its own graph, on its own stream) and the video encoder's backward run CONCURRENTLY — as they do
        # inside the single backward graph of a 1-GPU job — and each gradient class's buckets leave as soon as its
        # graph is through: the text encoder's under the rest of the video backward, the late video stages'
        # (most of the encoder's parameters, little of its time) under the token-heavy early stages;
        # step() -> reducer.finish() sends what is left (early stages + patch embedding: a few MB).
        main = torch.cuda.current_stream()
        ts = self._bwd_text_stream if self.graph_bwd_text is not None else None
        self.reducer.launch_where(self._ready('h'))
        if ts is not None:
            ts.wait_stream(main)
            with torch.cuda.stream(ts):
                self.graph_bwd_text.replay()
                self.reducer.launch_where(self._ready('h', 't'))
        for i, g in enumerate(self.graph_bwd_video):
            g.replay()
            if i + 1 < len(self.graph_bwd_video):          # came down to the in-encoder cut
                self.reducer.launch_where(self._ready('h', 'v1') if ts is not None else self._ready('h', 't', 'v1'))
        if ts is not None:
            main.wait_stream(ts)

    def capture(self, batch, warmup=2):
        """Capture the rank-local part of the step — CloverPretrain.encode and its backward, ~2300 of
        the ~2400 kernel launches — as two hipGraphs sharing one memory pool (static shapes; no
        data-dependent shape and no host sync by construction).  The cross-rank part (feature
        all-gather, InfoNCE/rank losses, logged-scalar all-reduce) stays eager between the two
        replays, so no RCCL call is ever captured and the same code path serves 1..8 GPUs."""
        model = self.model
        # the recognizer names the batch entries its rank-local part consumes (CloverPretrain: captions + both
        # masks; CloverFinetune retrieval: captions only) and may return no rank-local loss (mlm is None)
        aux = {k: None for k in getattr(model, 'CLV_ENCODE_KEYS', ('token_ids', 'input_mask', 'mlm_label',
                                                                  'v_token_mask'))}
        self._static_batch = {k: v.clone() for k, v in batch.items()}
        sb = self._static_batch

        import inspect
        cut_ok = self.reducer.active and hasattr(model, 'backbone')
        text_ok = (cut_ok and hasattr(model, 'text_backbone')
                   and 'text_cut' in inspect.signature(model.encode).parameters
                   and os.environ.get('CLOVER_TEXT_CUT', '1') == '1')
        if text_ok and getattr(self, '_bwd_text_stream', None) is None:
            self._bwd_text_stream = torch.cuda.Stream()

        def encode(vcuts=None, tcuts=None):
            kw = {k: sb[k] for k in aux}
            if text_ok:
                kw['text_cut'] = tcuts
            return model.encode(sb['imgs'], video_cut=vcuts, **kw)

        def roots(emb, mlm, demb, dmlm):
            return [emb] + ([mlm] if mlm is not None else []), [demb] + ([dmlm] if mlm is not None else [])

        def bwd_cut(cuts):
            self._backward(lambda: torch.autograd.backward([o for o, _ in cuts], [leaf.grad for _, leaf in cuts]))

        def bwd_video(cuts):                   # output cut first, then the in-encoder cut (backward order)
            for c in reversed(cuts):
                bwd_cut([c])
        self.reducer.enabled = False           # no collective may be issued from inside a capture; in graph
        self.reducer.reset()                   # mode the engine launches the buckets between / after the replays
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                vcuts = [] if cut_ok else None
                tcuts = [] if text_ok else None
                self._ft.done.clear()
                emb, mlm = encode(vcuts, tcuts)
                self._backward(lambda: torch.autograd.backward(*roots(emb, mlm, torch.zeros_like(emb),
                                                                      torch.zeros_like(mlm) if mlm is not None else None)))
                if vcuts:
                    bwd_video(vcuts)
                if tcuts:
                    bwd_cut(tcuts)
        torch.cuda.current_stream().wait_stream(side)
        # no autograd graph may survive into the capture: a live one pins the parameters'
        # AccumulateGrad nodes to the warm-up stream and their accumulation escapes the hipGraph
        del emb, mlm, vcuts, tcuts
        for seg in self.segments:
            seg.flat_g.zero_()
        self.sumsq.zero_()                    # (the warm-up backwards left norm partial sums: fused-norm mode)
        if self._norm_tables is not None:
            self._norm_state.dirty = False
        self._ft.done.clear()                 # the captured backward holds the first-touch stores of a fresh step
        torch.cuda.synchronize()
        gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        gb2 = None
        gb3 = torch.cuda.CUDAGraph() if text_ok else None
        # thread_local: RCCL's watchdog thread (W > 1) keeps polling events while we capture
        vcuts = [] if cut_ok else None
        tcuts = [] if text_ok else None
        with torch.cuda.graph(gf, capture_error_mode='thread_local'):
            emb, mlm = encode(vcuts, tcuts)
        self._static_demb = torch.zeros_like(emb)
        self._static_dmlm = torch.zeros_like(mlm) if mlm is not None else None
        # Data-parallel jobs: the bf16 wire copy of every bucket is written INSIDE the backward graph that completes its
        # gradients (the pack kernels are plain launches; only the RCCL calls stay between the replays)
        pack = cut_ok and os.environ.get('CLOVER_PACK_IN_GRAPH', '1') == '1'
        prepacked = set()
        with torch.cuda.graph(gb, pool=gf.pool(), capture_error_mode='thread_local'):
            self._backward(lambda: torch.autograd.backward(*roots(emb, mlm, self._static_demb, self._static_dmlm)))
            if pack:
                prepacked |= self.reducer.pack_where(self._ready('h'))
        if cut_ok:
            gb2 = []
            ncut = len(vcuts)
            for i, c in enumerate(reversed(vcuts)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=gf.pool(), capture_error_mode='thread_local'):
                    bwd_cut([c])
                    if pack:
                        # as _replay_backward launches them: the late video stages after the first cut graph, what is
                        # left of the video encoder (and, without a text cut, of the text encoder) after the last
                        cls = ('v1',) if i + 1 < ncut else (('v0', 'v1') if text_ok else ('v0', 'v1', 't'))
                        prepacked |= self.reducer.pack_where(self._ready(*cls), skip=prepacked)
                gb2.append(g)
        if text_ok:
            # captured LAST and into a pool of its own: it replays concurrently with the video graph, so neither
            # may recycle memory the other still reads (blocks this capture frees return to gf's pool after the
            # video graph has been laid out; its own temporaries never alias the video graph's)
            with torch.cuda.graph(gb3, capture_error_mode='thread_local'):
                bwd_cut(tcuts)
                if pack:
                    prepacked |= self.reducer.pack_where(self._ready('t'), skip=prepacked)
        self._prepacked = frozenset(prepacked)
        self._stale_views = self._stale_sinks()    # first-touch slots this geometry's backward never writes
        for seg in self.segments:
            seg.flat_g.zero_()                     # a capture pass does not execute kernels; be explicit
        self.sumsq.zero_()
        if self._norm_tables is not None and self._norm_state.dirty:
            raise RuntimeError('fused gradient norm: the captured backward gives a first-touch sink two gradients; build the '
                               'engine with CLOVER_FUSED_NORM=0')
        self._ft.done.clear()
        self.graph, self.graph_bwd, self.graph_bwd_video, self.graph_bwd_text = gf, gb, gb2, gb3
        self._static_emb, self._static_mlm = emb, mlm
        self.graph_loss, self._loss_io = self._capture_loss_graph()
        self._static_cuts = (vcuts, tcuts)
        self._active_sig = self._signature(batch)
        self._captures[self._active_sig] = tuple(getattr(self, f) for f in self._CAPTURE_FIELDS)
        return True

    def _capture_loss_graph(self):
        """The section between the two encode graphs — contrastive / rank losses on the gathered embeddings, the loss
        sum and its backward down to d emb / d mlm — as a hipGraph over a static gathered tensor.  The all-gather that
        fills it and the all-reduce of the logged scalars stay eager (no RCCL call is ever captured); the gather's
        backward is the local slice (gather_loss.py:64-72), taken inside the graph."""
        import inspect
        model = self.model
        if (os.environ.get('CLOVER_LOSS_GRAPH', '0') != '1' or not hasattr(model, 'contrastive_losses')
                or 'gathered' not in inspect.signature(model.contrastive_losses).parameters
                or not getattr(getattr(model, 'ssl_loss', None), 'equal_batch', False)):
            return None, None
        emb = self._static_emb
        B = emb.shape[0]
        g_static = torch.zeros((self.world * B,) + tuple(emb.shape[1:]), device=emb.device, dtype=torch.float32)
        g_static[self.rank * B:(self.rank + 1) * B].copy_(emb)
        out = {}

        def run():
            g = g_static.detach().requires_grad_()
            mlm = self._static_mlm.detach().requires_grad_() if self._static_mlm is not None else None
            losses = model.contrastive_losses(None, mlm, gathered=g)
            loss, names, packed = self._parse_root_scaled(losses, reduce=False)
            loss.backward(gradient=self._root_gradient(loss))
            self._static_demb.copy_(g.grad[self.rank * B:(self.rank + 1) * B])
            if mlm is not None:
                self._static_dmlm.copy_(mlm.grad)
            out.update(names=names, packed=packed, loss=loss.detach())

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # a pool of its own: it is captured after the backward graphs but replays BEFORE them — in the shared pool its
        # temporaries would land on blocks the backward capture released, i.e. on the activations the backward still reads
        gl = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gl, capture_error_mode='thread_local'):
            run()
        return gl, (g_static, out['names'], out['packed'], out['loss'])

    def optimizer_step(self):
        """Global-norm clip + AdamW on the slabs; no host synchronisation.  The clip coefficient, Adam's bias
        corrections and Adam's own step count live on the device (``clv_optim_prep``): a step with a non-finite
        gradient norm is skipped there and does not advance Adam's count (mmcv_Fp16OptimizerHook.py:123-141), while
        the LR index moves on like the reference's ``runner.iter``."""
        lr = self.current_lr()
        self.lr_iter += 1
        self.step_count += 1
        if self.graph is None and not self._stale_cleared:
            # a caller's own loss.backward() (the style of an OptimizerHook) that did not go through finish_backward():
            # first-touch slots this backward did not reach still hold the previous step's gradient — clear them before
            # the norm.  With a reduced wire copy it is too late for that (the stale values have been exchanged).
            stale = self._stale_sinks()
            if stale:
                if self.wire is not None and self.reducer.active:
                    raise RuntimeError('first-touch gradient slots were not reached by this backward and the gradient '
                                       'exchange has already run: call engine.finish_backward() instead of '
                                       'engine.reducer.finish()')
                torch._foreach_zero_(stale)
        self._stale_cleared = False
        gscale = 1.0 / self.world      # DDP averages the summed gradients (the loss scale is divided out on the device)
        grads = self.wire if self.wire is not None else [seg.flat_g for seg in self.segments]   # reduced gradients
        fused = self._norm_tables is not None
        if fused and self._norm_state.dirty:   # a sink got a second gradient this step (ops.linear_wgrad): recompute it all
            if self.graph is not None:
                raise RuntimeError('fused gradient norm: a first-touch sink received two gradients inside a captured step; '
                                   'build the engine with CLOVER_FUSED_NORM=0')
            self._norm_state.dirty = False
            self.sumsq.zero_()
            fused = False
        if fused:                              # the first-touch slots' sums sit in the norm slots already
            for seg, (tab, nblk) in zip(self.segments, self._norm_tables):
                ops.sumsq_ranges(seg.flat_g, tab, nblk, self.sumsq)
        else:
            for g in grads:
                ops.sumsq_accumulate(g, self.sumsq)
        ops.optim_prep(self.sumsq, self.optim_state, self.betas[0], self.betas[1],
                       self.grad_clip if self.grad_clip else 0.0, gscale,
                       slots=self.sumsq[16:] if self._norm_tables is not None else None)
        for seg, g in zip(self.segments, grads):
            ops.adamw_step_dev(seg.flat_p, g, seg.exp_avg, seg.exp_avg_sq, seg.shadow, self.optim_state,
                               lr * seg.lr_mult, self.betas[0], self.betas[1], self.eps, seg.weight_decay)
            seg.refresh_transposed()
        self.zero_grads()
        self.last_lr = lr

    def optimizer_state(self):
        """AdamW state for checkpoints: per segment the flat moments + names/offsets, and the step count."""
        st = ops.optim_state_read(self.optim_state)
        return dict(step=st['t'], skipped=st['skipped'], calls=self.step_count, lr_iter=self.lr_iter,
                    loss_scaler=self.loss_scaler_state(st),
                    segments=[dict(names=list(sg.names), offsets=list(sg.offsets), weight_decay=sg.weight_decay,
                                   lr_mult=sg.lr_mult,
                                   exp_avg=sg.exp_avg.detach().cpu(), exp_avg_sq=sg.exp_avg_sq.detach().cpu())
                              for sg in self.segments])

    def load_optimizer_state(self, state):
        self.step_count = int(state.get('calls', state['step']))
        self.lr_iter = int(state.get('lr_iter', state['step']))
        self.optim_state.zero_()
        self.optim_state[5] = int(state['step'])                 # Adam's t (ops.optim_state_read layout)
        self.optim_state[6] = int(state.get('skipped', 0))
        if self._scaler_on:
            sc = dict(cur_scale=self.scaler_cfg['init_scale'], cur_iter=0, last_overflow_iter=-1)
            if self.scaler_cfg['mode'] == 'dynamic':      # (a static scale is configuration, not state)
                sc.update(state.get('loss_scaler') or {})
            ops.optim_state_set_scaler(self.optim_state, sc['cur_scale'], self.scaler_cfg['mode'] == 'dynamic',
                                       self.scaler_cfg['scale_factor'], self.scaler_cfg['scale_window'],
                                       sc['cur_iter'], sc['last_overflow_iter'])
        assert len(self.segments) == len(state['segments']), 'optimizer state belongs to a different parameter layout'
        for sg, st in zip(self.segments, state['segments']):
            assert list(sg.names) == list(st['names']), 'optimizer state belongs to a different parameter layout'
            for key in ('weight_decay', 'lr_mult'):          # the slab's options are baked into its AdamW launch
                if key in st and abs(float(st[key]) - float(getattr(sg, key))) > 1e-12:
                    raise ValueError(f'optimizer state was saved with {key}={st[key]} for the slab of {sg.names[0]} ... '
                                     f'but the current paramwise_cfg gives {getattr(sg, key)}')
            if 'offsets' not in st or list(st['offsets']) == list(sg.offsets):
                sg.exp_avg.copy_(st['exp_avg'])
                sg.exp_avg_sq.copy_(st['exp_avg_sq'])
            else:
                # a checkpoint written under another slab layout (slot alignment / phantom padding rows changed between
                # rounds): same parameters, other offsets — re-scatter the moments per parameter by name instead of copying
                # the flat buffers (which would fail on the size, or silently misalign the moments)
                old_off = list(st['offsets'])
                sg.exp_avg.zero_()
                sg.exp_avg_sq.zero_()
                for i, (name, p_) in enumerate(zip(sg.names, sg.params)):
                    n = p_.numel()
                    if old_off[i] + n > st['exp_avg'].numel() or (i + 1 < len(old_off) and old_off[i + 1] - old_off[i] < n):
                        raise ValueError(f'optimizer state: the saved slot of {name} holds fewer than {n} elements '
                                         f'(checkpoint from a different model?)')
                    for dst, key in ((sg.exp_avg, 'exp_avg'), (sg.exp_avg_sq, 'exp_avg_sq')):
                        dst[sg.offsets[i]:sg.offsets[i] + n].copy_(st[key][old_off[i]:old_off[i] + n])
        self.refresh_shadow()

    def refresh_shadow(self):
        """Re-derive the bf16 compute copy from the fp32 slabs — after anything that writes parameters behind the
        optimizer's back (``load_state_dict`` of a checkpoint: it copies into the slab views in place)."""
        for sg in self.segments:
            sg.shadow.copy_(sg.flat_p)
            sg.refresh_transposed()

    @property
    def loss_scale(self):
        """The scale the NEXT backward will use (host sync; 1.0 without a scaler)."""
        return ops.optim_state_read(self.optim_state)['loss_scale'] if self._scaler_on else 1.0

    def loss_scaler_state(self, st=None):
        """LossScaler.state_dict() (fp16_utils.py:364-373) — what the reference keeps in runner.meta['fp16']['loss_scaler']
        (mmcv_Fp16OptimizerHook.py:147-149); None without a scaler."""
        if not self._scaler_on:
            return None
        st = st or ops.optim_state_read(self.optim_state)
        return dict(cur_scale=st['loss_scale'], cur_iter=st['scale_iter'], mode=self.scaler_cfg['mode'],
                    last_overflow_iter=st['last_overflow'], scale_factor=st['scale_factor'],
                    scale_window=st['scale_window'])

    def grad_norm(self):
        """Global gradient norm of the last step's (averaged) gradients — host sync, logging only."""
        return ops.optim_state_read(self.optim_state)['norm']

    def adam_steps(self):
        """Adam's own step count = optimizer steps actually taken (host sync)."""
        return ops.optim_state_read(self.optim_state)['t']
