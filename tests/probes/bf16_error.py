"""Probe: bf16 product vs fp32 oracle loss error under realistic init (not a test)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tests/golden'); sys.path.insert(0, ROOT + '/tests')
import torch
import closed_form as cf
import clover_amd
from oracle import model as om

torch.manual_seed(0)
cfg = cf.tiny_model_cfg()
m = clover_amd.build_model(cfg)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
sd = {k: v.clone() for k, v in m.state_dict().items() if 'relative_position_index' not in k}
if scale != 1.0:
    for k, v in sd.items():
        if v.dim() >= 2:
            v.mul_(scale)
    m.load_state_dict(sd, strict=False)
m = m.cuda().eval()
for B in [2, 4, 8]:
    batch = cf.cf_batch(B, tag=f'p{B}')
    batch['imgs'] = torch.randn(batch['imgs'].shape)
    out = m.train_step({k: v.cuda() for k, v in batch.items()}, None)
    ref = om.forward_train(sd, batch, cf.oracle_cfg_from(cfg), gather=False)
    _, lv = om.parse_losses(ref)
    print(B, {k: (round(lv[k], 5), round(out['log_vars'][k] - lv[k], 5)) for k in lv})
