"""Runner + config layer under a tools/train.py-style driver, restated from the reference:

* ``Config``: python config files with ``_base_`` inheritance, attribute access and ``--cfg-options a.b=c``
  overrides (what ``mmcv.Config.fromfile`` / ``merge_from_dict`` give ``tools/train.py:261-263``; mmcv is not
  installed here);
* ``scaled_lr``: the linear scaling rule of ``tools/train.py:160-166``;
* ``CloverRunner``: epoch loop, hook call points and the multi-dataloader interleave of
  ``mmaction/core/runner/clover_runner.py:17-35,60-96`` (one optimizer step per loader per batch index), including
  its behaviour once the shorter loader is exhausted;
* checkpoints in the reference's ``{'meta', 'state_dict', 'optimizer'}`` layout (``epoch_based_runner.py:25-58``).

The step itself is ``CloverEngine.step`` (or any object with ``train_step``); nothing here touches the GPU.
"""
import ast
import copy
import os
import runpy
import time
from itertools import zip_longest

import torch


# --------------------------------------------------------------------------- config
class ConfigDict(dict):
    """dict with attribute access (nested dicts are wrapped on read)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            v = ConfigDict(v)
            self[k] = v                                  # wrap in place: mutations through the view must stick
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _merge(base, child):
    """mmcv semantics: dicts merge recursively, anything else is replaced; ``_delete_=True`` in the child dict
    drops the base dict instead of merging into it."""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = copy.deepcopy({kk: vv for kk, vv in v.items() if kk != '_delete_'} if isinstance(v, dict) else v)
    return out


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg', ConfigDict(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def _load(path):
        ns = runpy.run_path(path)
        cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v)
               and not isinstance(v, type(os))}
        bases = cfg.pop('_base_', [])
        bases = [bases] if isinstance(bases, str) else list(bases)
        merged = {}
        for b in bases:
            merged = _merge(merged, Config._load(os.path.join(os.path.dirname(path), b)))
        return _merge(merged, cfg)

    @classmethod
    def fromfile(cls, path):
        return cls(cls._load(os.path.abspath(path)), filename=path)

    def merge_from_dict(self, options):
        """``{'model.backbone.depths': [2, 2]}`` style overrides (tools/train.py:263)."""
        for key, val in (options or {}).items():
            d = self._cfg
            parts = key.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = val

    def __getattr__(self, k):
        return getattr(self._cfg, k)

    def __setattr__(self, k, v):
        self._cfg[k] = v

    def get(self, k, default=None):
        return self._cfg.get(k, default)

    def setdefault(self, k, v):
        return self._cfg.setdefault(k, v)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg))


def parse_cfg_options(items):
    """['a.b=1', 'c=[1,2]', 'd=text'] -> dict (mmcv DictAction)."""
    out = {}
    for it in items or []:
        k, v = it.split('=', 1)
        try:
            out[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            out[k] = v
    return out


def scaled_lr(cfg, world_size):
    """tools/train.py:160-166 — if the optimizer config carries ``base_lr`` it is REMOVED and
    ``lr = base_lr * videos_per_gpu * world_size`` is set, with videos_per_gpu read from the TOP level of the
    config (``cfg.get('videos_per_gpu', 1)``, not ``cfg.data``), exactly as the reference does."""
    opt = cfg.optimizer
    if 'base_lr' in opt:
        base_lr = opt.pop('base_lr')
        opt['lr'] = base_lr * cfg.get('videos_per_gpu', 1) * world_size
    return opt.get('lr')


# --------------------------------------------------------------------------- runner
class Hook:
    """The six call points the reference's runners use (mmcv Hook names)."""

    def before_run(self, runner): pass
    def after_run(self, runner): pass
    def before_train_epoch(self, runner): pass
    def after_train_epoch(self, runner): pass
    def before_train_iter(self, runner): pass
    def after_train_iter(self, runner): pass


class LrUpdaterHook(Hook):
    """mmcv 1.3.x ``CosineAnnealingLrUpdaterHook`` with ``by_epoch=False`` and linear warm-up (the reference's
    ``lr_config``, pretrain_webvid_cc3m.py:139-140).  The schedule is indexed by ``runner.iter`` —
    which the multi-loader runner advances once per batch INDEX, so both loaders' steps of one index share one LR
    (clover_runner.py:76-91) — against ``runner._max_iters`` (epochs x the LONGEST loader), and ``warmup_iters`` given
    in epochs is multiplied by that loader's length in ``before_train_epoch`` as mmcv does.  ``before_train_iter``
    hands the value to the stepper (``set_lr``), i.e. before the step that uses it, starting at iter 0."""

    def __init__(self, base_lr, min_lr_ratio=None, min_lr=None, warmup=None, warmup_iters=0, warmup_ratio=0.1,
                 warmup_by_epoch=False, **_ignored):
        assert (min_lr is None) ^ (min_lr_ratio is None), 'exactly one of min_lr / min_lr_ratio'
        assert warmup in (None, 'linear'), 'only linear warm-up is restated'
        self.base_lr = base_lr
        self.min_lr_ratio = min_lr_ratio if min_lr_ratio is not None else min_lr / base_lr
        self.warmup, self.warmup_ratio = warmup, warmup_ratio
        self.warmup_epochs = warmup_iters if warmup_by_epoch else None
        self.warmup_iters = None if warmup_by_epoch else warmup_iters
        self.history = []

    def before_train_epoch(self, runner):
        if self.warmup_iters is None:
            self.warmup_iters = self.warmup_epochs * runner.epoch_len

    def lr_at(self, it, max_iters):
        from .engine import cosine_lr
        return cosine_lr(self.base_lr, it, max_iters, self.min_lr_ratio,
                         self.warmup_iters if self.warmup else 0, self.warmup_ratio)

    def before_train_iter(self, runner):
        lr = self.lr_at(runner.iter, runner._max_iters)
        self.history.append(lr)
        if hasattr(runner.stepper, 'set_lr'):
            runner.stepper.set_lr(lr)


class LogHook(Hook):
    """TextLoggerHook stand-in: keeps the last ``log_vars`` (same keys the reference logs) every `interval` iters."""

    def __init__(self, interval=10, printer=None):
        self.interval, self.printer, self.records = interval, printer, []

    def after_train_iter(self, runner):
        if runner.inner_iter % self.interval == 0 and runner.outputs is not None:
            rec = dict(epoch=runner.epoch + 1, iter=runner.inner_iter + 1,
                       **{k: float(v) for k, v in dict(runner.outputs['log_vars']).items()})
            self.records.append(rec)
            if self.printer:
                self.printer(rec)


class CheckpointHook(Hook):
    def __init__(self, out_dir, interval=1):
        self.out_dir, self.interval = out_dir, interval

    def after_train_epoch(self, runner):
        if (runner.epoch + 1) % self.interval == 0:
            runner.save_checkpoint(self.out_dir, f'epoch_{runner.epoch + 1}.pth')


class CloverRunner:
    """``stepper`` is a CloverEngine (``step(batch)``) or a module with ``train_step(batch, optimizer)``."""

    def __init__(self, stepper, model=None, optimizer=None, work_dir=None, max_epochs=None, meta=None):
        self.stepper = stepper
        self.model = model if model is not None else getattr(stepper, 'model', stepper)
        self.optimizer = optimizer
        self.work_dir, self.meta = work_dir, meta or {}
        self._max_epochs, self._max_iters = max_epochs, None
        self.epoch, self.iter, self.inner_iter = 0, 0, 0
        self.hooks, self.outputs, self.mode = [], None, None
        self.epoch_len = None                 # len(runner.data_loader): the longest loader in multi-loader mode

    def register_hook(self, hook):
        self.hooks.append(hook)

    def call_hook(self, name):
        for h in self.hooks:
            getattr(h, name)(self)

    def run_iter(self, data_batch):
        if hasattr(self.stepper, 'step'):
            self.outputs = self.stepper.step(data_batch)
        else:
            self.outputs = self.stepper.train_step(data_batch, self.optimizer)
        if not isinstance(self.outputs, dict):
            raise TypeError('train_step must return a dict')            # epoch_based_runner run_iter

    # ---- clover_runner.py:17-35 (single loader)
    def _train_single(self, loader):
        self.epoch_len = len(loader)
        self._max_iters = self._max_epochs * len(loader)
        self.call_hook('before_train_epoch')
        for i, data_batch in enumerate(loader):
            self.inner_iter = i
            self.call_hook('before_train_iter')
            self.run_iter(data_batch)
            self.call_hook('after_train_iter')
            self.iter += 1
            if i >= len(loader) - 1:
                break
        self.call_hook('after_train_epoch')
        self.epoch += 1

    # ---- clover_runner.py:60-96 (several loaders; one optimizer step per loader per batch index)
    def _train_multi(self, loaders):
        self.epoch_len = max(len(ld) for ld in loaders)                # :62-69 — data_loader = the longest one
        self._max_iters = self._max_epochs * self.epoch_len
        self.call_hook('before_train_epoch')
        short_loader = None
        for batch_idx, batches in enumerate(zip_longest(*loaders)):
            self.inner_iter = batch_idx
            for loader_idx, data_batch in enumerate(batches):
                # :78-82 — the FIRST exhausted loader is restarted once, and from then on EVERY slot of the row
                # (also the longer loader's, whose own batch is dropped) is fed from that restarted iterator
                if short_loader is None and data_batch is None:
                    short_loader = iter(loaders[loader_idx])
                    data_batch = next(short_loader)
                elif short_loader is not None:
                    data_batch = next(short_loader)
                self.call_hook('before_train_iter')
                self.run_iter(data_batch)
                self.call_hook('after_train_iter')
            self.iter += 1                                   # :91 — counts batch indices, not optimizer steps
            if batch_idx >= max(len(ld) - 1 for ld in loaders):
                break
        self.call_hook('after_train_epoch')
        self.epoch += 1

    def train(self, data_loader, multi=False):
        """One epoch over a loader, or (multi=True) over a list of loaders interleaved."""
        self.model.train()
        self.mode = 'train'
        if multi:
            self._train_multi(list(data_loader))
        else:
            self._train_single(data_loader)

    def run(self, data_loaders, workflow=(('train', 1),), max_epochs=None):
        """clover_runner.py:98-163: workflow of ('train', n) phases until max_epochs; a list of loaders with a
        single ('train', n) entry is the multi-dataset mode."""
        if max_epochs is not None:
            self._max_epochs = max_epochs
        assert self._max_epochs is not None, 'max_epochs must be specified'
        self.call_hook('before_run')
        while self.epoch < self._max_epochs:
            for i, (mode, epochs) in enumerate(workflow):
                if mode != 'train':
                    raise ValueError(f'runner has no method named "{mode}" to run an epoch')
                for _ in range(epochs):
                    if self.epoch >= self._max_epochs:
                        break
                    multi = len(workflow) == 1 and len(data_loaders) > 1
                    self.train(data_loaders if multi else data_loaders[i], multi=multi)
        self.call_hook('after_run')

    # ---- checkpoints (epoch_based_runner.py:25-58 layout)
    def save_checkpoint(self, out_dir, filename):
        os.makedirs(out_dir, exist_ok=True)
        meta = dict(self.meta, epoch=self.epoch + 1, iter=self.iter, time=time.asctime())
        sd = {k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        ckpt = dict(meta=meta, state_dict=sd)
        if hasattr(self.stepper, 'optimizer_state'):
            ckpt['optimizer'] = self.stepper.optimizer_state()
        elif self.optimizer is not None:
            ckpt['optimizer'] = self.optimizer.state_dict()
        path = os.path.join(out_dir, filename)
        torch.save(ckpt, path)
        return path

    def load_checkpoint(self, path, strict=False):
        ckpt = torch.load(path, map_location='cpu')
        sd = ckpt.get('state_dict', ckpt)
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}        # DDP prefix
        res = self.model.load_state_dict(sd, strict=strict)
        if hasattr(self.stepper, 'refresh_shadow'):
            self.stepper.refresh_shadow()                # the engine computes from a bf16 copy of the weights
        return ckpt, res

    def resume(self, path):
        ckpt, _ = self.load_checkpoint(path)
        self.epoch = ckpt['meta'].get('epoch', 0)
        self.iter = ckpt['meta'].get('iter', 0)
        if 'optimizer' in ckpt and hasattr(self.stepper, 'load_optimizer_state'):
            self.stepper.load_optimizer_state(ckpt['optimizer'])
        return ckpt
