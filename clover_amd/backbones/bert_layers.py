"""BERT building blocks with HuggingFace's module/parameter names, MI355X-native compute.

The reference delegates these to the un-vendored ``transformers==4.6.1`` (install.sh:25;
call sites bert_from_hugface.py:13-15,30, cross_transformer.py:24-29,109-110,
mlm_itm_head.py:33-41).  Semantics restated from that version: embeddings = word +
token_type(0) + absolute position -> LayerNorm(eps) -> dropout; each layer = post-LN
self-attention (scores/sqrt(d) + additive mask (1-m)*-10000, softmax, dropout, context) ->
dense + dropout + residual + LN -> GELU(erf) FFN -> dense + dropout + residual + LN.

Q/K/V keep their three separate Parameters (checkpoint compatibility) but run as ONE
[3H,H] GEMM; attention is the fused HIP kernel (``clv_attn_fwd`` mode 0); residual adds are
fused into the LayerNorm kernel.
"""
import json
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..nn import LayerNorm, Linear, to_bf16

BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12,
                 hidden_act='gelu')


def resolve_bert_config(pretrained_model='bert-base-uncased', bert_config=None, **overrides):
    """Config resolution without network: explicit ``bert_config`` dict > a local directory
    ``<pretrained_model>/config.json`` (what the reference's from_pretrained reads offline) >
    the built-in bert-base-uncased table."""
    cfg = dict(BERT_BASE)
    if bert_config is not None:
        cfg.update(bert_config)
    elif pretrained_model and os.path.isfile(os.path.join(str(pretrained_model), 'config.json')):
        with open(os.path.join(str(pretrained_model), 'config.json')) as f:
            cfg.update({k: v for k, v in json.load(f).items() if k in cfg})
    elif pretrained_model not in (None, 'bert-base-uncased'):
        raise FileNotFoundError(f'no local BERT config for {pretrained_model!r} (no network access); '
                                'pass bert_config=dict(...) or a directory with config.json')
    cfg.update({k: v for k, v in overrides.items() if v is not None})
    if cfg['hidden_act'] != 'gelu':
        raise NotImplementedError('only erf-GELU BERT is supported')
    return cfg


def hf_legacy_key(k):
    """Key renames ``from_pretrained`` applies while loading (transformers 4.6.1 ``modeling_utils.py``
    ``_load_state_dict_into_model``): the stock bert-base-uncased weights still carry the TF names
    ``LayerNorm.gamma`` / ``LayerNorm.beta``."""
    if 'gamma' in k:
        k = k.replace('gamma', 'weight')
    if 'beta' in k:
        k = k.replace('beta', 'bias')
    return k


def load_pretrained_dir(module, pretrained_model, prefix='', allow_unexpected=(), allow_missing=()):
    """Load ``<dir>/model.safetensors`` / ``pytorch_model.bin`` into ``module`` the way the reference's
    ``from_pretrained`` calls do (bert_from_hugface.py:13-15, cross_transformer.py:24-29, mlm_itm_head.py:33-35):
    legacy ``gamma`` / ``beta`` names are mapped to ``weight`` / ``bias``, the MLM head's output bias is stored as
    ``cls.predictions.bias`` (HF ties ``decoder.bias`` to it) and its ``decoder.weight`` is tied to the word
    embeddings when the file does not carry it.  Returns False when the directory holds no weight file.
    Raises if a parameter of ``module`` is left un-initialised (except names matching ``allow_missing``) — a silent
    partial load would start pre-training from a half-random text encoder."""
    d = str(pretrained_model)
    sd = None
    if os.path.isfile(os.path.join(d, 'model.safetensors')):
        from safetensors.torch import load_file
        sd = load_file(os.path.join(d, 'model.safetensors'))
    elif os.path.isfile(os.path.join(d, 'pytorch_model.bin')):
        sd = torch.load(os.path.join(d, 'pytorch_model.bin'), map_location='cpu')
    if sd is None:
        return False
    full = {hf_legacy_key(k): v for k, v in sd.items()}
    if prefix.startswith('bert.') and not any(k.startswith('bert.') for k in full):
        prefix = prefix[len('bert.'):]               # a bare BertModel checkpoint (HF strips / adds base_model_prefix)
    sd = {k[len(prefix):]: v for k, v in full.items() if k.startswith(prefix)} if prefix else dict(full)
    if not sd:
        import warnings
        warnings.warn(f'{d}: no weights under {prefix!r}; {type(module).__name__} keeps its fresh initialisation '
                      '(what from_pretrained does, with the same warning)')
        return False
    if prefix == 'cls.predictions.':
        if 'bias' in sd:
            sd.setdefault('decoder.bias', sd['bias'])
            del sd['bias']
        for tied in ('bert.embeddings.word_embeddings.weight', 'embeddings.word_embeddings.weight'):
            if 'decoder.weight' not in sd and tied in full:
                sd['decoder.weight'] = full[tied]
    res = module.load_state_dict(sd, strict=False)
    missing = [k for k in res.missing_keys if not any(a in k for a in allow_missing)]
    unexpected = [k for k in res.unexpected_keys
                  if not any(a in k for a in tuple(allow_unexpected) + ('position_ids',))]
    if missing:
        raise RuntimeError(f'{d}: {len(missing)} parameter(s) of {type(module).__name__} not found in the '
                           f'checkpoint (prefix {prefix!r}): {missing[:8]}')
    if unexpected:
        import warnings
        warnings.warn(f'{d}: {len(unexpected)} checkpoint key(s) under {prefix!r} have no counterpart in '
                      f'{type(module).__name__}: {unexpected[:8]}')
    return True


def init_bert_weights(module, std=0.02):
    """HF BertPreTrainedModel._init_weights."""
    for m in module.modules():
        if isinstance(m, nn.Linear):
            m.weight.data.normal_(mean=0.0, std=std)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.Embedding):
            m.weight.data.normal_(mean=0.0, std=std)
        elif isinstance(m, nn.LayerNorm):
            m.bias.data.zero_()
            m.weight.data.fill_(1.0)


class BertEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.word_embeddings = nn.Embedding(cfg['vocab_size'], cfg['hidden_size'], padding_idx=0)
        self.position_embeddings = nn.Embedding(cfg['max_position_embeddings'], cfg['hidden_size'])
        self.token_type_embeddings = nn.Embedding(cfg['type_vocab_size'], cfg['hidden_size'])
        self.LayerNorm = LayerNorm(cfg['hidden_size'], eps=cfg['layer_norm_eps'])
        self.dropout = nn.Dropout(cfg['hidden_dropout_prob'])

    def forward(self, input_ids, past_key_values_length=0):
        L = input_ids.shape[1]
        pos = torch.arange(past_key_values_length, past_key_values_length + L, device=input_ids.device)
        we = self.word_embeddings
        word = ops.embedding(input_ids, we.weight, we.padding_idx) if input_ids.is_cuda else we(input_ids)
        e = word + self.token_type_embeddings.weight[0] + self.position_embeddings(pos)[None]
        return self.dropout(self.LayerNorm(to_bf16(e)))


class BertSelfAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H = cfg['hidden_size']
        self.num_attention_heads = cfg['num_attention_heads']
        self.query = Linear(H, H)
        self.key = Linear(H, H)
        self.value = Linear(H, H)
        self.attn_dropout_p = cfg['attention_probs_dropout_prob']

    def clv_fuse_groups(self):
        """Parameters applied as ONE GEMM: the engine lays them out adjacently and hands back fused
        views (``_clv_fused``), so the Q|K|V projection needs no per-step cat / cast / gradient split."""
        return [[self.query.weight, self.key.weight, self.value.weight],
                [self.query.bias, self.key.bias, self.value.bias]]

    def forward(self, x, kmask):
        fused = getattr(self, '_clv_fused', None)
        if fused is not None:
            w, b = fused
        else:
            w = torch.cat([self.query.weight, self.key.weight, self.value.weight], dim=0)
            b = torch.cat([self.query.bias, self.key.bias, self.value.bias], dim=0)
        qkv = ops.linear(x, w, b)
        return ops.seq_attention(qkv.contiguous(), kmask, self.num_attention_heads,
                                 self.attn_dropout_p if self.training else 0.0)


class BertSelfOutput(nn.Module):
    def __init__(self, cfg, in_features=None):
        super().__init__()
        H = cfg['hidden_size']
        self.dense = Linear(in_features or H, H)
        self.LayerNorm = LayerNorm(H, eps=cfg['layer_norm_eps'])
        self.dropout = nn.Dropout(cfg['hidden_dropout_prob'])

    def forward(self, hidden, residual, fork=False):
        """LayerNorm(dropout(dense(h)) + residual): the dropout runs inside the LayerNorm kernel.  fork=True
        returns the output twice (one storage): one edge for the next sub-layer's GEMM, one for its residual
        input, and the two gradients meet inside the LayerNorm backward kernel instead of an autograd add."""
        return self.LayerNorm(self.dense(hidden), residual=residual,
                              x_dropout_p=self.dropout.p if self.training else 0.0, fork=fork)


class BertAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = BertSelfAttention(cfg)
        self.output = BertSelfOutput(cfg)

    def forward(self, x, kmask, x_res=None, fork=False):
        return self.output(self.self(x, kmask), x if x_res is None else x_res, fork=fork)


class BertIntermediate(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.dense = Linear(cfg['hidden_size'], cfg['intermediate_size'])

    def forward(self, x):
        return ops.gelu(self.dense(x))


class BertLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = BertAttention(cfg)
        self.intermediate = BertIntermediate(cfg)
        self.output = BertSelfOutput(cfg, in_features=cfg['intermediate_size'])

    def forward(self, x, kmask, x_res=None, fork_out=False):
        """x_res: the alias of x a forked producer handed out for the residual input (see BertSelfOutput)."""
        a, a_res = self.attention(x, kmask, x_res, fork=True)
        fc1, out = self.intermediate.dense, self.output
        if ops.mlp_gelu_ok(a, fc1.out_features):
            # FFN with the activation inside the GEMMs (clv_gemm_nt epilogues: bias + GELU keeping GELU' forward, one
            # multiply in the input gradient of the second GEMM) — no standalone GELU pass over the [tokens, 3072] tensor
            h = ops.mlp_gelu(a, fc1.weight, fc1.bias, out.dense.weight, out.dense.bias)
            return out.LayerNorm(h, residual=a_res, x_dropout_p=out.dropout.p if self.training else 0.0, fork=fork_out)
        return self.output(self.intermediate(a), a_res, fork=fork_out)


class BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([BertLayer(cfg) for _ in range(cfg['num_hidden_layers'])])

    def forward(self, x, kmask):
        x_res, last = None, len(self.layer) - 1
        for i, layer in enumerate(self.layer):
            if i < last:
                x, x_res = layer(x, kmask, x_res, fork_out=True)
            else:
                x = layer(x, kmask, x_res)
        return x


class BertPooler(nn.Module):
    """Present for state_dict compatibility; statically unused on the pre-training path
    (only last_hidden_state is read, multimodal_transformer_pretrain.py:101,111)."""

    def __init__(self, cfg):
        super().__init__()
        self.dense = Linear(cfg['hidden_size'], cfg['hidden_size'])


class BertModel(nn.Module):
    def __init__(self, cfg, add_pooling_layer=True):
        super().__init__()
        self.config = cfg
        self.embeddings = BertEmbeddings(cfg)
        self.encoder = BertEncoder(cfg)
        self.pooler = BertPooler(cfg) if add_pooling_layer else None

    def forward(self, input_ids=None, attention_mask=None):
        kmask = extended_attention_mask(attention_mask)
        # flush_point: the encoder's backward ends here — its deferred weight gradients leave on the stream it ran on
        h = self.encoder(ops.flush_point(self.embeddings(input_ids)), kmask)
        return {'last_hidden_state': h}


def extended_attention_mask(mask):
    """transformers 4.6.1 get_extended_attention_mask for a 2-D mask, as the [B,S] fp32 additive
    key mask the attention kernel takes: (1 - mask) * -10000.0."""
    return ((1.0 - mask.to(torch.float32)) * -10000.0).contiguous()
