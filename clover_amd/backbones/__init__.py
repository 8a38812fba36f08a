from .bert_from_hugface import BertFromPretrained
from .cross_transformer import CrossModalTransformerFromPretrained
from .swin_transformer_3d import SwinTransformer3D

__all__ = ['SwinTransformer3D', 'BertFromPretrained', 'CrossModalTransformerFromPretrained']
