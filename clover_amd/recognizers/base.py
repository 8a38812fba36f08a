"""``BaseRecognizer`` — the train_step / forward / _parse_losses contract of
mmaction/models/recognizers/base.py:16-372 that mmcv's ``EpochBasedRunner.run_iter`` drives
(``outputs = model.train_step(data_batch, optimizer)``)."""
from abc import ABCMeta, abstractmethod
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from .. import builder
from ..utils.dist import collectives_active


class LazyLogVars(OrderedDict):
    """``log_vars`` whose values are fetched from the device on first access.

    The reference does one all-reduce + ``.item()`` host sync PER logged key every step
    (base.py:281-286).  Here the six scalars travel as one packed tensor (one collective) and
    the host copy happens only when somebody reads them (the text logger does every 20
    iterations, configs/_base_/default_runtime.py:2-7); reading gives plain floats, as before."""

    def __init__(self, names, packed):
        super().__init__()
        self._packed = packed
        self._names = list(names)
        for n in self._names:
            OrderedDict.__setitem__(self, n, None)
        self._resolved = False

    def fresh(self):
        """A new unresolved view of the same device tensor (after a hipGraph replay refreshed it)."""
        return LazyLogVars(self._names, self._packed)

    def _resolve(self):
        if not self._resolved:
            vals = self._packed.detach().float().cpu().tolist()
            for n, v in zip(self._names, vals):
                OrderedDict.__setitem__(self, n, v)
            self._resolved = True

    def __getitem__(self, k):
        self._resolve()
        return OrderedDict.__getitem__(self, k)

    def items(self):
        self._resolve()
        return OrderedDict.items(self)

    def values(self):
        self._resolve()
        return OrderedDict.values(self)

    def get(self, k, default=None):
        self._resolve()
        return OrderedDict.get(self, k, default)


class BaseRecognizer(nn.Module, metaclass=ABCMeta):
    def __init__(self, backbone, cls_head=None, neck=None, freeze_stage=None, freeze_except=[], train_cfg=None,
                 test_cfg=None):
        super().__init__()
        self.backbone_from = 'mmaction2'
        btype = backbone['type'] if isinstance(backbone, dict) else ''
        if isinstance(btype, str) and btype.split('.')[0] in ('mmcls', 'torchvision', 'timm'):
            raise NotImplementedError(f'{btype}: only registry backbones are on the MI355X path')
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        if cls_head:
            raise NotImplementedError('cls_head is not part of the pre-training path (configs set cls_head=None)')
        self.cls_head = None
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.aux_info = []
        if train_cfg is not None and 'aux_info' in train_cfg:
            self.aux_info = train_cfg['aux_info']
        self.max_testing_views = None
        if test_cfg is not None and 'max_testing_views' in test_cfg:
            self.max_testing_views = test_cfg['max_testing_views']
            assert isinstance(self.max_testing_views, int)
        self.feature_extraction = bool(test_cfg and test_cfg.get('feature_extraction', False))
        self.blending = None
        if train_cfg is not None and 'blending' in train_cfg:
            raise NotImplementedError('mini-batch blending is not used by the Clover configs')
        self.fp16_enabled = False
        self.lazy_log_vars = True
        self.init_weights()
        if freeze_stage is not None:
            self._freeze(freeze_stage=freeze_stage, freeze_except=freeze_except)

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_cls_head(self):
        return hasattr(self, 'cls_head') and self.cls_head is not None

    def _freeze(self, freeze_stage, freeze_except):
        """Name-substring freezing (reference :138-163)."""
        freeze_norm_layer = 'norm_layer' not in freeze_except
        norm_types = (nn.modules.batchnorm._BatchNorm, nn.modules.instancenorm._InstanceNorm, nn.LayerNorm,
                      nn.GroupNorm)
        for n, m in self.named_modules():
            if any(en in n for en in freeze_except):
                continue
            for fn in freeze_stage:
                if fn in n:
                    if isinstance(m, norm_types):
                        if not freeze_norm_layer:
                            break
                        m.eval()
                    for p in m.parameters():
                        p.requires_grad = False
                    break

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_neck:
            self.neck.init_weights()

    def extract_feat(self, imgs):
        return self.backbone(imgs)

    @abstractmethod
    def forward_train(self, imgs, labels, **kwargs):
        """Defines the computation performed at every call when training."""

    @abstractmethod
    def forward_test(self, imgs):
        """Defines the computation performed at every call when evaluation and testing."""

    def _parse_losses(self, losses, reduce=True):
        """loss = sum of every entry whose key contains 'loss'; log_vars = all entries (+ 'loss'),
        averaged over ranks (reference :254-288)."""
        vals = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                vals[name] = value if value.dim() == 0 else value.mean()       # mean of a scalar is the scalar
            elif isinstance(value, list):
                vals[name] = sum(_l.mean() for _l in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        # one stack + one sum instead of a chain of scalar kernels (and of their backward nodes)
        names = list(vals.keys())
        stacked = torch.stack([v.float().reshape(()) for v in vals.values()])
        pick = [i for i, k in enumerate(names) if 'loss' in k]
        loss = stacked.sum() if len(pick) == len(names) else stacked[pick].sum()
        names.append('loss')
        packed = torch.cat([stacked.detach(), loss.detach().reshape(1)])
        # loss scale of the 16-bit backward (fp16 build: 1024; bf16: 1): applied to the gradient at the root, divided out of
        # every parameter gradient again — by the engine's optimizer kernels, or by the hooks below for plain autograd
        from .. import _lib, ops
        dev_scale = getattr(self, '_clv_loss_scale_dev', None)      # an engine's device-resident scaler (engine.CloverEngine)
        if dev_scale is False or getattr(self, '_clv_root_scaled', False):
            pass          # an engine that runs unscaled (loss_scale=1), or one that passes the scale as the root gradient
        elif dev_scale is not None and loss.requires_grad:
            loss = ops.scale_grad_dev(loss, dev_scale)
        elif _lib.LOSS_SCALE != 1.0 and loss.requires_grad:
            BaseRecognizer._register_unscale_hooks(self, _lib.LOSS_SCALE)     # (unbound: toy recognizers borrow _parse_losses)
            loss = ops.scale_grad(loss, _lib.LOSS_SCALE)
        if not reduce:                                   # inside a hipGraph capture: the caller averages over the ranks
            return loss, names, packed
        if collectives_active():
            packed = packed / dist.get_world_size()
            dist.all_reduce(packed)
        if getattr(self, 'lazy_log_vars', True):
            return loss, LazyLogVars(names, packed)
        return loss, OrderedDict(zip(names, packed.cpu().tolist()))

    def _register_unscale_hooks(self, scale):
        """Parameters whose gradient arrives through autograd (no engine sink) get it divided by the loss scale as it is
        accumulated, so that ``loss.backward()`` leaves true-scale ``.grad`` tensors.  An engine removes the hook of every
        parameter it takes over (their gradients live, scaled, in its slabs: engine._Segment)."""
        if getattr(self, '_clv_unscale_done', None) == scale:
            return
        from .. import ops
        for q in self.parameters():
            if q.requires_grad and getattr(q, '_clv_unscale', None) is None and getattr(q, '_clv_grad', None) is None:
                q._clv_unscale = q.register_hook(ops.unscale_hook)     # divides by the scale of the backward that is running
        object.__setattr__(self, '_clv_unscale_done', scale)

    def forward(self, imgs=None, label=None, return_loss=True, **kwargs):
        if kwargs.get('gradcam', False):
            raise NotImplementedError('gradcam is outside the pre-training path')
        if return_loss:
            if label is None:
                raise ValueError('Label should not be None.')
            return self.forward_train(imgs, label, **kwargs)
        return self.forward_test(imgs, **kwargs)

    def train_step(self, data_batch, optimizer=None, **kwargs):
        """-> dict(loss, log_vars, num_samples) (reference :304-347)."""
        imgs = data_batch['imgs']
        label = data_batch['label']
        aux_info = {}
        for item in self.aux_info:
            assert item in data_batch
            aux_info[item] = data_batch[item]
        losses = self(imgs, label, return_loss=True, **aux_info)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(next(iter(data_batch.values()))))

    def val_step(self, data_batch, optimizer=None, **kwargs):
        return self.train_step(data_batch, optimizer, **kwargs)
