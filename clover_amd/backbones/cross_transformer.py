"""Cross-modal fusion transformer, registered as ``CrossModalTransformerFromPretrained``
with the reference's kwargs and parameter names
(mmaction/models/backbones/cross_transformer.py:12-124)."""
import torch
import torch.nn as nn

from ..builder import BACKBONES
from ..nn import LayerNorm, Linear, to_bf16, trunc_normal_
from .bert_layers import (BertEmbeddings, BertEncoder, extended_attention_mask, init_bert_weights,
                          load_pretrained_dir, resolve_bert_config)


@BACKBONES.register_module()
class CrossModalTransformerFromPretrained(nn.Module):
    def __init__(self, pretrained_model='bert-base-uncased', img_in_size=768, hidden_size=768, num_frames=4,
                 spacial_tokens=7 * 7, token_types=2, num_hidden_layers=12, layer_norm_eps=1e-12,
                 word_pos_start=False, use_prompt=False, use_text_cls=False, return_mask=False, bert_config=None,
                 **kwargs):
        super().__init__()
        cfg = resolve_bert_config(pretrained_model, bert_config, layer_norm_eps=layer_norm_eps,
                                  num_hidden_layers=num_hidden_layers)
        assert cfg['hidden_size'] == hidden_size, 'hidden_size must match the BERT config'
        self.bert_embedding = BertEmbeddings(cfg)     # unused when text_input_embeds is given (pre-training)
        self.bert_encoder = BertEncoder(cfg)
        init_bert_weights(self.bert_embedding)
        init_bert_weights(self.bert_encoder)
        load_pretrained_dir(self.bert_embedding, pretrained_model, prefix='bert.embeddings.')
        # the fusion encoder takes the first num_hidden_layers layers of the checkpoint (cross_transformer.py:24-29)
        load_pretrained_dir(self.bert_encoder, pretrained_model, prefix='bert.encoder.',
                            allow_unexpected=tuple(f'layer.{i}.' for i in range(cfg['num_hidden_layers'], 64)))
        self.use_prompt = use_prompt
        if not use_text_cls:
            self.all_cls_token = nn.Parameter(torch.zeros(1, 1, hidden_size))
            trunc_normal_(self.all_cls_token, mean=0., std=.02)
            if self.use_prompt:
                self.prompt_token = nn.Parameter(torch.zeros(1, 4, hidden_size))
                trunc_normal_(self.prompt_token, mean=0., std=.02)
        else:
            self.all_cls_token = None
        self.vis_space_pos = nn.Parameter(0.02 * torch.randn(1, 1, spacial_tokens, hidden_size))
        self.vis_tempor_pos = nn.Parameter(0.02 * torch.randn(1, num_frames, 1, hidden_size))
        self.token_type_embeddings = nn.Embedding(token_types, hidden_size)
        self.norm = LayerNorm(hidden_size)
        self.word_pos_start = word_pos_start
        self.num_frames = num_frames
        self.spacial_tokens = spacial_tokens
        self.img_in_size = img_in_size
        self.hidden_size = hidden_size
        self.return_mask = return_mask
        if img_in_size != hidden_size:
            self.fc_in = Linear(img_in_size, hidden_size)
        self.fp16_enabled = False

    def _text_and_pos(self, text_embeddings, T, S, D):
        tt = self.token_type_embeddings.weight
        # the two embedding additions as single bf16 kernels (the sums are rounded to bf16 either way; the addends'
        # own rounding is 2^-9 of a ~0.02-sized embedding)
        text = to_bf16(text_embeddings) + to_bf16(tt[1])
        pos = (self.vis_space_pos + self.vis_tempor_pos[:, :T, :, :]).reshape(1, T * S, D) + tt[0]
        return text, to_bf16(pos)

    def prepare(self, text_input_embeds, text_input_mask, B, T, S):
        """Everything of forward() that does not depend on the visual tokens — text + token-type embedding, the visual
        position table, the multimodal key mask — so that a caller can run these ~10 launch-bound kernels early, on the
        text encoder's stream, instead of between the video encoder and the first fusion layer."""
        D = self.hidden_size
        text = text_input_embeds
        mask = text_input_mask
        if text.shape[0] != B:
            text = text.view(B, -1, text.shape[-1])
            mask = mask.view(B, -1)
        text, pos = self._text_and_pos(text, T, S, D)
        n_vis = T * S + (1 if self.all_cls_token is not None else 0)
        mm_mask = torch.cat([torch.ones(B, n_vis, dtype=mask.dtype, device=text.device), mask], dim=1)
        return dict(key=(B, T, S), text=text, pos=pos, mask=mm_mask, ext_mask=extended_attention_mask(mm_mask))

    def forward(self, visual_token=None, text_input_ids=None, text_input_mask=None, text_input_embeds=None, **kwargs):
        """visual_token [B,T,S,Din]; returns mapping with last/t_/v_last_hidden_state (reference :64-124)."""
        if self.img_in_size != self.hidden_size:
            visual_token = self.fc_in(visual_token)
        B, T, S, D = visual_token.shape
        p_k_v_l = T * S + 1 if self.word_pos_start else 0
        if text_input_embeds is None:
            text_embeddings = self.bert_embedding(text_input_ids, past_key_values_length=p_k_v_l)
        else:
            text_embeddings = text_input_embeds
        if text_embeddings.shape[0] != B:
            text_embeddings = text_embeddings.view(B, -1, text_embeddings.shape[-1])
            text_input_mask = text_input_mask.view(B, -1)
        prepared = kwargs.get('prepared')
        if prepared is not None and prepared['key'] == (B, T, S) and not self.use_prompt:
            text_embeddings, pos, mm_mask_p, ext_mask_p = (prepared['text'], prepared['pos'], prepared['mask'],
                                                           prepared['ext_mask'])
        else:
            text_embeddings, pos = self._text_and_pos(text_embeddings, T, S, D)
            mm_mask_p = ext_mask_p = None
        visual = self.norm(to_bf16(visual_token.reshape(B, T * S, D)) + pos)
        if self.use_prompt:
            visual = torch.cat([visual, to_bf16(self.prompt_token).expand(B, -1, -1),
                                to_bf16(self.all_cls_token).expand(B, -1, -1)], dim=1)
            n_vis = T * S + 5
        elif self.all_cls_token is not None:
            visual = torch.cat([visual, to_bf16(self.all_cls_token).expand(B, -1, -1)], dim=1)
            n_vis = T * S + 1
        else:
            n_vis = T * S
        feat = torch.cat([visual, text_embeddings], dim=1)
        if mm_mask_p is not None:
            mm_mask, ext_mask = mm_mask_p, ext_mask_p
        else:
            mm_mask = torch.cat([torch.ones(B, n_vis, dtype=text_input_mask.dtype, device=feat.device),
                                 text_input_mask], dim=1)
            ext_mask = extended_attention_mask(mm_mask)
        h = self.bert_encoder(feat, ext_mask)
        out = {'last_hidden_state': h,
               't_last_hidden_state': h[:, n_vis:],
               'v_last_hidden_state': h[:, :T * S]}
        if self.all_cls_token is not None:
            out['cls_last_hidden_state'] = h[:, n_vis - 1:n_vis]
        if self.return_mask:
            return out, mm_mask
        return out
