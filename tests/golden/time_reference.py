"""BUILD-CONTAINER ONLY: seconds per training iteration (forward_train + _parse_losses + backward, fp32, all host cores) of the
REAL reference and of the oracle restatement on the same weights / batch / cores — the ratio bench.py's `cpu_baseline`
(kind "port": the oracle) has to be read with.
    python tests/golden/time_reference.py T 8     |     python tests/golden/time_reference.py B 32"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import closed_form as cf  # noqa: E402,F401
import ref_harness as H  # noqa: E402

variant, frames = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
import bench  # noqa: E402
import clover_amd  # noqa: E402
from oracle import model as om  # noqa: E402

SCRATCH = '/tmp/clover_golden_scratch_base'
H.make_bert_dir(SCRATCH, hidden=768, layers=12, heads=12, inter=3072, vocab=30522, max_pos=512)
H.init_dist_single()
cores = len(os.sched_getaffinity(0))
torch.set_num_threads(cores)
cfg = bench.model_cfg(variant, frames)
torch.manual_seed(4321)
own = clover_amd.build_model(cfg).eval()
sd = {k: v.detach().clone() for k, v in own.state_dict().items()}
del own
batch = bench.synthetic_batch(2, frames, 32, seed=77)

rcfg = bench.model_cfg(variant, frames)
for part in ('mm_backbone', 'text_backbone'):
    rcfg[part].pop('bert_config', None)
m = H.build_reference_model(rcfg, SCRATCH)
m.load_state_dict({k: v for k, v in sd.items() if k in m.state_dict()}, strict=False)
m.eval()


def ref_iter():
    m.zero_grad()
    aux = {k: batch[k] for k in ('token_ids', 'segment_ids', 'input_mask', 'mlm_label', 'v_token_mask')}
    losses = m(batch['imgs'], batch['label'], return_loss=True, **aux)
    loss, _ = m._parse_losses(losses)
    loss.backward()


P = {k: v.detach().float().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()
     if 'relative_position_index' not in k}


def oracle_iter():
    for v in P.values():
        v.grad = None
    loss, _ = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
    loss.backward()


out = {}
for name, fn in (('reference', ref_iter), ('oracle', oracle_iter)):
    fn()                                                     # warm-up
    t0 = time.time()
    for _ in range(iters):
        fn()
    out[name] = (time.time() - t0) / iters
print(f'Swin-{variant} {frames}f B=2 on {cores} cores: reference {out["reference"]:.2f} s/iter ({2 / out["reference"]:.3f} pairs/s), '
      f'oracle {out["oracle"]:.2f} s/iter ({2 / out["oracle"]:.3f} pairs/s), oracle / reference speed = {out["reference"] / out["oracle"]:.2f}')
