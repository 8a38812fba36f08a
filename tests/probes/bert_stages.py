"""Probe: per-stage error of the product BERT (bf16) vs the fp32 oracle on closed-form weights."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests/golden'); sys.path.insert(0, ROOT + '/tests')
import torch
import closed_form as cf, gutil
import clover_amd
from oracle import model as om

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max()).item(), ((a - b).norm() / b.norm()).item()

cfg = cf.tiny_model_cfg()
m = clover_amd.build_model(cfg)
sd = cf.cf_state(gutil.manifest())
m.load_state_dict(sd, strict=False)
m = m.cuda().eval()
b = cf.cf_batch(3, tag='bf')
ids, mask = b['token_ids'][:, 0], b['input_mask'][:, 0]
ocfg = cf.oracle_cfg_from(cfg)
P = sd
bert = m.text_backbone.bert
with torch.no_grad():
    e = bert.embeddings(ids.cuda())
    e_ref = om.bert_embeddings(P, 'text_backbone.bert.embeddings.', ids, 1e-12)
    print('embeddings', rel(e, e_ref))
    from clover_amd.backbones.bert_layers import extended_attention_mask
    km = extended_attention_mask(mask.cuda())
    h, h_ref = e, e_ref
    ext = om.extended_mask(mask)
    for i, layer in enumerate(bert.encoder.layer):
        pre = f'text_backbone.bert.encoder.layer.{i}.'
        # attention sub-block on the REFERENCE input (isolates the stage)
        a = layer.attention(h_ref.cuda().to(torch.bfloat16), km)
        # oracle pieces
        B, S, Hd = h_ref.shape
        heads = 2; hd = Hd // heads
        def split(t): return t.view(B, S, heads, hd).permute(0, 2, 1, 3)
        q = split(om.linear(P, pre + 'attention.self.query', h_ref)); k = split(om.linear(P, pre + 'attention.self.key', h_ref)); v = split(om.linear(P, pre + 'attention.self.value', h_ref))
        sc = q @ k.transpose(-1, -2) / hd ** 0.5 + ext
        print(f'layer {i}: |scores| max', sc[sc > -1000].abs().max().item(), ' q max', q.abs().max().item())
        ctx = (sc.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, S, Hd)
        a_ref = om.layer_norm(P, pre + 'attention.output.LayerNorm', om.linear(P, pre + 'attention.output.dense', ctx) + h_ref, 1e-12)
        ctx_p = layer.attention.self(h_ref.cuda().to(torch.bfloat16), km)
        print(f'layer {i}: ctx', rel(ctx_p, ctx), ' attn-out', rel(a, a_ref))
        out_ref = om.bert_layer(P, pre, h_ref, ext, heads, 1e-12)
        out = layer(h_ref.cuda().to(torch.bfloat16), km)
        print(f'layer {i}: layer-out (ref input)', rel(out, out_ref))
        h = layer(h, km)
        h_ref = out_ref
        print(f'layer {i}: cumulative', rel(h, h_ref))
