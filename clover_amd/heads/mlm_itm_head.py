"""Masked-LM head, registered as ``MLMHead`` (mmaction/models/heads/mlm_itm_head.py:25-52):
HF ``cls.predictions`` = transform (dense, GELU, LayerNorm 1e-12) + decoder Linear(H, V).
The reference copies the weights out of ``BertForMaskedLM.from_pretrained('bert-base-uncased')``
(:33-35, its own copy — not tied to the text encoder); here they are built from
(hidden_size, vocab_size) and loaded from a checkpoint / local BERT directory when present."""
import os

import torch.nn as nn

from .. import ops
from ..backbones.bert_layers import init_bert_weights, load_pretrained_dir
from ..builder import HEADS
from ..nn import LayerNorm, Linear


class BertPredictionHeadTransform(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.dense = Linear(hidden_size, hidden_size)
        self.LayerNorm = LayerNorm(hidden_size, eps=1e-12)
        self.fp16_enabled = False

    def forward(self, hidden_states):
        return self.LayerNorm(ops.gelu(self.dense(hidden_states)))


class BertLMPredictionHead(nn.Module):
    def __init__(self, hidden_size, vocab_size):
        super().__init__()
        self.transform = BertPredictionHeadTransform(hidden_size)
        self.decoder = Linear(hidden_size, vocab_size, bias=True)
        if vocab_size % 8:
            # the decoder's GEMMs run on the step's own kernels with the vocabulary padded by phantom rows the engine appends
            # to the slab slots (engine._Segment) — to a multiple of 64: 16-byte row groups for the scores, whole 64-deep
            # stages for the input gradient, which contracts over the vocabulary and takes the transposed bf16 shadow
            pad = -vocab_size % 64
            self.decoder.weight._clv_pad_rows = pad
            self.decoder.bias._clv_pad_rows = pad
            self.decoder.weight._clv_want_t = True
        init_bert_weights(self)
        if os.path.isdir('bert-base-uncased'):
            load_pretrained_dir(self, 'bert-base-uncased', prefix='cls.predictions.')
        self.fp16_enabled = False

    def forward(self, hidden_states):
        h = self.transform(hidden_states)
        if ops.mlm_decoder_ok(h, self.decoder.weight, self.decoder.bias):
            return ops.mlm_decoder(h, self.decoder.weight, self.decoder.bias)
        return self.decoder(h)


@HEADS.register_module()
class MLMHead(nn.Module):
    def __init__(self, hidden_size, vocab_size):
        super().__init__()
        self.predictions = BertLMPredictionHead(hidden_size, vocab_size)
        self.fp16_enabled = False

    def forward(self, sequence_output):
        """[B,L,H] -> prediction scores [B,L,V] (bf16)."""
        return self.predictions(sequence_output)
