"""Registries and build functions with the reference's names
(mmaction/models/builder.py:8-86): one MODELS registry aliased as BACKBONES / HEADS /
RECOGNIZERS / LOSSES, ``build_backbone/head/loss/recognizer/model(cfg)``."""
import warnings

from .registry import Registry

MODELS = Registry('models')
BACKBONES = MODELS
NECKS = MODELS
HEADS = MODELS
RECOGNIZERS = MODELS
LOSSES = MODELS
LOCALIZERS = MODELS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_recognizer(cfg, train_cfg=None, test_cfg=None):
    if train_cfg is not None or test_cfg is not None:
        warnings.warn('train_cfg and test_cfg is deprecated, please specify them in model', UserWarning)
    assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg specified in both outer field and model field'
    assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg specified in both outer field and model field '
    return RECOGNIZERS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    args = dict(cfg)
    obj_type = args.pop('type')
    if obj_type in RECOGNIZERS:
        return build_recognizer(cfg, train_cfg, test_cfg)
    raise ValueError(f'{obj_type} is not registered in LOCALIZERS, RECOGNIZERS or DETECTORS')


def register_into_mmcv():
    """When real mmcv is present, expose the classes under mmcv.cnn.MODELS as well, so the
    reference's tools/train.py -> build_model(cfg.model) resolves to these modules."""
    try:
        from mmcv.cnn import MODELS as MMCV_MODELS  # noqa: N811
    except Exception:
        return False
    for name, cls in MODELS.module_dict.items():
        MMCV_MODELS.register_module(name=name, force=True, module=cls)
    return True
