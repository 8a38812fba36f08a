import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests'); sys.path.insert(0, ROOT + '/tests/golden')
import torch, closed_form as cf, gutil, clover_amd
from clover_amd.engine import CloverEngine, paramwise_weight_decay
DEV = 'cuda:0'
def make_model():
    m = clover_amd.build_model(cf.tiny_model_cfg()); m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False); return m.to(DEV).eval()
b = {k: v.to(DEV) for k, v in cf.cf_batch(2, tag='eng').items()}
m1, m2 = make_model(), make_model()
eng = CloverEngine(m1, b, lr=1e-3, weight_decay=0.005, grad_clip=15.0, max_iters=10 ** 9)
wd = paramwise_weight_decay(m2, 0.005, 0.0, 0.0, {'relative_position_bias_table': dict(decay_mult=0.)})
named = [(n, p) for n, p in m2.named_parameters() if n not in eng.unused_names]
opt = torch.optim.AdamW([dict(params=[p], weight_decay=wd[n]) for n, p in named], lr=1e-3, betas=(0.9, 0.98), eps=1e-8)
for it in range(3):
    eng.step(b)
    opt.zero_grad(set_to_none=True)
    m2.train_step(b, None)['loss'].backward()
    torch.nn.utils.clip_grad_norm_([p for _, p in named], 15.0)
    for g in opt.param_groups: g['lr'] = 1e-3 * (1e-3 + 0.5 * (1 - 1e-3) * (1 + 1.0))
    opt.step()
p1 = dict(m1.named_parameters()); res = []
for n, p in named:
    d = (p1[n].detach() - p.detach()).abs().max().item(); res.append((d / (p.detach().abs().max().item() + 1e-6), n, d, p.detach().abs().max().item()))
for r in sorted(res, reverse=True)[:6]: print(r)
