from .base import BaseRecognizer
from .multimodal_transformer_finetune import CloverFinetune
from .multimodal_transformer_pretrain import CloverPretrain

__all__ = ['BaseRecognizer', 'CloverPretrain', 'CloverFinetune']
