import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests/golden'); sys.path.insert(0, ROOT + '/tests')
import torch, closed_form as cf, gutil, clover_amd
from clover_amd.engine import CloverEngine
m = clover_amd.build_model(cf.tiny_model_cfg()); m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False)
m = m.cuda().eval()
b = {k: v.cuda() for k, v in cf.cf_batch(4, tag='eng4').items()}
eng = CloverEngine(m, b, lr=2e-4, weight_decay=0.0, grad_clip=15.0, max_iters=10**9)
eng.step(b)
print('captured', eng.capture(b))
seg = eng.segments[0]
for i in range(3):
    p0 = seg.flat_p.clone(); s0 = seg.shadow.float().clone()
    for k, v in eng._static_batch.items():
        pass
    eng.graph.replay()
    torch.cuda.synchronize()
    print('after replay: |g|', seg.flat_g.norm().item(), 'loss', float(eng._static_out['loss']), 'packed', eng._static_out['log_vars']._packed.tolist()[-1])
    eng.reducer.finish(); eng.optimizer_step(); torch.cuda.synchronize()
    print('  after opt: |dp|', (seg.flat_p - p0).norm().item(), '|dshadow|', (seg.shadow.float() - s0).norm().item(), 'sumsq', eng.sumsq.item())
print('---- which grads are non-finite after a replay')
for seg in eng.segments:
    seg.flat_g.zero_()
eng.graph.replay(); torch.cuda.synchronize()
bad = []
for seg in eng.segments:
    for n, p in zip(seg.names, seg.params):
        if not torch.isfinite(p.grad).all():
            bad.append((n, int((~torch.isfinite(p.grad)).sum()), p.grad.numel()))
print(len(bad), 'bad of', sum(len(s.names) for s in eng.segments))
for x in bad[:40]: print(x)
