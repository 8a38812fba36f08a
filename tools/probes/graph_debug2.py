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
def bad():
    out = []
    for seg in eng.segments:
        for n, p in zip(seg.names, seg.params):
            if not torch.isfinite(p.grad).all():
                out.append(n)
    return out
for i in range(3):
    print('pre-replay |g| per seg', [float(s.flat_g.abs().max()) for s in eng.segments])
    eng.graph.replay(); torch.cuda.synchronize()
    print(i, 'bad:', bad()[:8], 'loss', float(eng._static_out['loss']))
    eng.optimizer_step(); torch.cuda.synchronize()
    print('   param finite:', all(bool(torch.isfinite(s.flat_p).all()) for s in eng.segments), 'shadow finite', all(bool(torch.isfinite(s.shadow.float()).all()) for s in eng.segments))
