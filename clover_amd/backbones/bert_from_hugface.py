"""Text encoder, registered as ``BertFromPretrained`` with the reference's kwargs
(mmaction/models/backbones/bert_from_hugface.py:8-33)."""
import torch.nn as nn

from ..builder import BACKBONES
from .bert_layers import BertModel, init_bert_weights, load_pretrained_dir, resolve_bert_config


@BACKBONES.register_module()
class BertFromPretrained(nn.Module):
    def __init__(self, pretrained_model='bert-base-uncased', layer_norm_eps=1e-12, num_hidden_layers=12,
                 bert_config=None, **kwargs):
        super().__init__()
        cfg = resolve_bert_config(pretrained_model, bert_config, layer_norm_eps=layer_norm_eps,
                                  num_hidden_layers=num_hidden_layers)
        self.bert = BertModel(cfg)
        init_bert_weights(self.bert)
        load_pretrained_dir(self.bert, pretrained_model, prefix='bert.', allow_missing=('pooler.',),   # unused here
                            allow_unexpected=tuple(f'layer.{i}.' for i in range(cfg['num_hidden_layers'], 64)))

    def forward(self, token_ids=None, input_mask=None, **kwargs):
        """-> mapping with 'last_hidden_state' [B,L,H] (reference :26-32)."""
        return self.bert(input_ids=token_ids, attention_mask=input_mask)
