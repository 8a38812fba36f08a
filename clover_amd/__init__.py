"""clover_amd — MI355X-native implementation of Clover's video-text pre-training hot path
behind the reference's mmaction2 Registry / backbone / head / recognizer plugin API.

Importing the package registers every module under the reference's names; the HIP extension
(libclover_hip.so) is loaded on first kernel call and its absence raises — no CPU fallback.
"""
from . import backbones, heads, losses, recognizers  # noqa: F401  (registration side effects)
from .builder import (BACKBONES, HEADS, LOSSES, MODELS, RECOGNIZERS, build_backbone, build_head, build_loss,
                      build_model, build_recognizer, register_into_mmcv)

register_into_mmcv()

__version__ = '0.1.0'
__all__ = ['MODELS', 'BACKBONES', 'HEADS', 'LOSSES', 'RECOGNIZERS', 'build_backbone', 'build_head', 'build_loss',
           'build_model', 'build_recognizer']
