"""Reference import harness — BUILD-CONTAINER ONLY (never runs on the GPU box).

Imports the *real* reference modules from /root/reference (read-only) so that
``make_goldens.py`` can run them and dump input/output vectors.  Nothing of the
reference's source travels: only the numeric fixtures written by
``make_goldens.py`` are committed.

What it provides (recipe: SURVEY.md Appendix D):
  1. stand-in ``mmcv`` / ``timm`` modules (the real packages are not installed);
  2. namespace packages for ``mmaction.*`` whose ``__path__`` points into the
     read-only tree, so the heavy ``__init__`` files are bypassed;
  3. transformers 4.6.1 ``get_extended_attention_mask`` semantics ((1-m)*-10000);
  4. a scratch ``bert-base-uncased`` directory with a seeded random-init
     ``BertForPreTraining`` so every ``from_pretrained`` resolves offline;
  5. a 1-rank gloo process group (reference defect R2);
  6. the R1 pre-hook on ``mlm_ssl_V_head`` (2-D CLS row -> [B,1,D]).
"""
import importlib.machinery
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None, is_package=True)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class _Registry:
    """30-line stand-in for mmcv.utils.Registry (register_module/build/in)."""

    def __init__(self, name, parent=None, **kw):
        self.name = name
        self._module_dict = {}

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        def _reg(cls):
            self._module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return _reg(module)
        return _reg

    def build(self, cfg, default_args=None):
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        typ = args.pop('type')
        cls = self._module_dict[typ] if isinstance(typ, str) else typ
        return cls(**args)


class _DropPath(nn.Module):
    """timm DropPath (stochastic depth per sample)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        return x * mask / keep


def _trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


def _passthrough_decorator(*dargs, **dkw):
    def deco(f):
        return f
    return deco


def install_shims():
    if 'mmaction' in sys.modules:
        return
    from torch.nn.modules.batchnorm import _BatchNorm
    from torch.nn.modules.instancenorm import _InstanceNorm
    models_registry = _Registry('models')

    def get_dist_info():
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
        return 0, 1

    def digit_version(v):
        out = []
        for p in v.split('+')[0].split('.'):
            out.append(int(''.join(c for c in p if c.isdigit()) or 0))
        return tuple(out)

    _mod('mmcv')
    _mod('mmcv.cnn', MODELS=models_registry)
    _mod('mmcv.utils', Registry=_Registry, digit_version=digit_version,
         TORCH_VERSION=torch.__version__, print_log=lambda *a, **k: None,
         _BatchNorm=_BatchNorm, _InstanceNorm=_InstanceNorm,
         get_logger=lambda *a, **k: None,
         build_from_cfg=lambda cfg, reg, default_args=None: reg.build(cfg, default_args))
    _mod('mmcv.runner', get_dist_info=get_dist_info,
         load_checkpoint=lambda *a, **k: None, load_state_dict=lambda *a, **k: None,
         force_fp32=_passthrough_decorator, auto_fp16=_passthrough_decorator)
    _mod('mmcv.runner.dist_utils', allreduce_grads=lambda *a, **k: None)
    _mod('timm')
    _mod('timm.models')
    _mod('timm.models.layers', DropPath=_DropPath, trunc_normal_=_trunc_normal_)

    R = REF_ROOT + '/mmaction'
    _pkg('mmaction', R)
    core = _pkg('mmaction.core', R + '/core')
    core.top_k_accuracy = lambda *a, **k: None
    core.mean_average_precision = lambda *a, **k: None
    _pkg('mmaction.core.hooks', R + '/core/hooks')
    _pkg('mmaction.models', R + '/models')
    _pkg('mmaction.models.utils', R + '/models/utils')

    def import_module_error_func(name):
        def deco(f):
            return f
        return deco
    import logging
    # mmaction/utils/__init__.py pulls mmcv-only modules; numpy_norm.py (normalize_fn, used by the retrieval
    # metrics) is self-contained and is loaded from the reference file itself
    import importlib.util
    spec = importlib.util.spec_from_file_location('mmaction.utils.numpy_norm', R + '/utils/numpy_norm.py')
    numpy_norm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(numpy_norm)
    _mod('mmaction.utils', get_root_logger=lambda *a, **k: logging.getLogger('ref'),
         import_module_error_func=import_module_error_func, normalize_fn=numpy_norm.normalize_fn)

    # transformers 4.6.1 extended-mask semantics (install.sh:25 pins 4.6.1)
    from transformers.modeling_utils import ModuleUtilsMixin

    def get_extended_attention_mask(self, attention_mask, input_shape=None, device=None, dtype=None):
        if attention_mask.dim() == 3:
            ext = attention_mask[:, None, :, :]
        else:
            ext = attention_mask[:, None, None, :]
        ext = ext.to(dtype=torch.float32)
        return (1.0 - ext) * -10000.0
    ModuleUtilsMixin.get_extended_attention_mask = get_extended_attention_mask


def make_bert_dir(scratch, hidden=128, layers=2, heads=2, inter=512, vocab=1024, max_pos=64, seed=0):
    """Seeded random-init BertForPreTraining saved as ./bert-base-uncased in `scratch`."""
    from transformers import BertConfig, BertForPreTraining
    os.makedirs(scratch, exist_ok=True)
    d = os.path.join(scratch, 'bert-base-uncased')
    cfg = BertConfig(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers,
                     num_attention_heads=heads, intermediate_size=inter,
                     max_position_embeddings=max_pos, type_vocab_size=2,
                     hidden_act='gelu', layer_norm_eps=1e-12)
    try:
        cfg._attn_implementation = 'eager'
    except Exception:
        pass
    torch.manual_seed(seed)
    m = BertForPreTraining(cfg)
    m.save_pretrained(d)
    return d


def init_dist_single(port=29581):
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(port))
        dist.init_process_group('gloo', rank=0, world_size=1)


def build_reference_model(cfg_dict, scratch):
    """chdir to scratch (holding bert-base-uncased/), import reference, build model."""
    os.environ['HF_HUB_OFFLINE'] = '1'
    os.environ['TRANSFORMERS_OFFLINE'] = '1'
    install_shims()
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        import mmaction.models.backbones  # noqa: F401
        import mmaction.models.heads.ssl_head  # noqa: F401
        import mmaction.models.heads.mlm_itm_head  # noqa: F401
        import mmaction.models.losses.contrastive_loss  # noqa: F401
        import mmaction.models.losses.focal_loss  # noqa: F401
        import mmaction.models.losses.cross_entropy_loss  # noqa: F401
        import mmaction.models.recognizers.multimodal_transformer_pretrain  # noqa: F401
        import mmaction.models.recognizers.multimodal_transformer_finetune  # noqa: F401
        from mmaction.models.builder import build_model
        model = build_model(cfg_dict)
    finally:
        os.chdir(cwd)
    # R1: NCEHeadForVision.forward does img.mean(dim=1) on a 2-D CLS row.
    if getattr(model, 'mlm_ssl_V_head', None) is not None:
        model.mlm_ssl_V_head.register_forward_pre_hook(lambda m, a: (a[0].unsqueeze(1),))
    return model
