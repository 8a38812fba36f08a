"""Error isolation for the 1e-3 loss target: per-loss |delta| of the step against the reference
goldens (config 1, B = 4) and against the oracle at BASELINE config 2's shapes (B = 2), for
  fast            the bf16 training path,
  parity          fp32 storage + fp32 arithmetic (clover_amd/parity.py),
  parity+<kind>   parity mode with ONE bf16 rounding source re-injected (act / stream / weight / prob / input),
  parity+all      all five re-injected (should land near `fast`).
Writes gpurun_out/parity_isolate.json and prints a markdown table (pasted into DESIGN.md §2).
    python tools/parity_isolate.py            # on the GPU box
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import torch  # noqa: E402

LOSS_KEYS = ['mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss']


def main():
    import bench
    import closed_form as cf
    import gutil
    import clover_amd
    from clover_amd import parity
    from oracle import model as om
    dev = 'cuda'
    variants = [('fast', None), ('parity', ())] + [(f'parity+{k}', (k,)) for k in parity.ROUND_KINDS] + \
               [('parity+all', parity.ROUND_KINDS)]
    out = {}

    def run(model, batch, ref, tag):
        rows = {}
        for name, rounds in variants:
            with torch.no_grad():
                if rounds is None:
                    lv = model.train_step(batch, None)['log_vars']
                else:
                    with parity.mode(round=rounds):
                        lv = model.train_step(batch, None)['log_vars']
            rows[name] = {k: abs(lv[k] - ref[k]) for k in LOSS_KEYS}
        out[tag] = rows

    # config 1: reference goldens
    m = clover_amd.build_model(cf.tiny_model_cfg())
    m.load_state_dict(cf.cf_state(gutil.manifest()), strict=False)
    m = m.to(dev).eval()
    g = gutil.load('g_step.npz')
    for B in (1, 2, 4):
        batch = {k: v.to(dev) for k, v in cf.cf_batch(B, tag=f'step{B}').items()}
        run(m, batch, {k: float(g[f'B{B}.{k}']) for k in LOSS_KEYS}, f'config1_B{B}_vs_reference')

    # config 2 shapes: oracle
    torch.manual_seed(4321)
    cfg = bench.model_cfg('T', 8)
    m = clover_amd.build_model(cfg).eval()
    P = {k: v.detach().float() for k, v in m.state_dict().items() if 'relative_position_index' not in k}
    batch = bench.synthetic_batch(2, 8, 32, seed=77)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.no_grad():
        _, ref = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
    m = m.to(dev)
    run(m, {k: v.to(dev) for k, v in batch.items()}, ref, 'config2_shapes_B2_vs_oracle')

    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'parity_isolate.json'), 'w'), indent=1)
    for tag, rows in out.items():
        print(f'\n### {tag}\n| variant | ' + ' | '.join(LOSS_KEYS) + ' |\n|---|' + '---|' * len(LOSS_KEYS))
        for name, e in rows.items():
            print(f'| {name} | ' + ' | '.join(f'{e[k]:.1e}' for k in LOSS_KEYS) + ' |')


if __name__ == '__main__':
    main()
