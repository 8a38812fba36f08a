import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench, clover_amd
from clover_amd.engine import CloverEngine
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
model = clover_amd.build_model(bench.model_cfg('T', 8)).to(dev); model.train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(8, 8, 32, 1000).items()}
eng = CloverEngine(model, batch, lr=1e-5, weight_decay=0.005, grad_clip=15.0, max_iters=100000)
print('first-touch params', eng.first_touch_params, 'of', eng.num_params, 'zero views', None if eng._zero_views is None else (len(eng._zero_views), sum(v.numel() for v in eng._zero_views)))
