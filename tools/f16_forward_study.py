"""What would f16 storage + f16 MFMA operands in the FORWARD do to the step's losses?  (the reference's own
arithmetic type is fp16, configs/exp_local/pretrain_webvid_cc3m.py:21; `v_mfma_f32_16x16x32_f16` runs at the bf16 rate.)
Not by building f16 kernels: by running the step in parity mode (fp32 storage + fp32 arithmetic on the HIP kernels) with ALL
five rounding sources of the training path — activations, the residual stream, GEMM weights, attention probabilities, the
clip operand — re-injected as roundings to bf16 (which reproduces the shipped bf16 path) or to f16.  |loss - oracle| per
loss at BASELINE config 2 / 4 / 5 shapes (B = 2, eval mode, the oracle on the host cores):

    fast                 the shipped bf16 kernels
    parity + all (bf16)  fp32 arithmetic, every tensor rounded to bf16 where the bf16 path rounds
    parity + all (f16)   the same with f16 roundings = the f16-forward estimate
    parity               no rounding (the 1e-3 row)

    python tools/f16_forward_study.py [T8 B16 B32]          # on the GPU box; writes gpurun_out/f16_forward_study.json
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
    import clover_amd
    from clover_amd import parity
    from oracle import model as om
    which = sys.argv[1:] or ['T8', 'B16', 'B32']
    out = {}
    for tag in which:
        variant, frames = tag[0], int(tag[1:])
        torch.manual_seed(4321)
        cfg = bench.model_cfg(variant, frames)
        m = clover_amd.build_model(cfg).eval()
        P = {k: v.detach().float() for k, v in m.state_dict().items() if 'relative_position_index' not in k}
        batch = bench.synthetic_batch(2, frames, 32, seed=77)
        torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
        with torch.no_grad():
            _, ref = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
        m = m.to('cuda')
        b = {k: v.to('cuda') for k, v in batch.items()}
        rows = {}
        with torch.no_grad():
            lv = m.train_step(b, None)['log_vars']
            rows['fast (bf16 kernels)'] = {k: abs(float(lv[k]) - float(ref[k])) for k in LOSS_KEYS}
            for name, dt in (('parity + all roundings to bf16', torch.bfloat16), ('parity + all roundings to f16', torch.float16)):
                with parity.mode(round=parity.ROUND_KINDS, dtype=dt):
                    lv = m.train_step(b, None)['log_vars']
                rows[name] = {k: abs(float(lv[k]) - float(ref[k])) for k in LOSS_KEYS}
            with parity.mode():
                lv = m.train_step(b, None)['log_vars']
            rows['parity (fp32)'] = {k: abs(float(lv[k]) - float(ref[k])) for k in LOSS_KEYS}
        out[f'Swin-{variant} {frames}f B=2'] = rows
        del m
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'f16_forward_study.json'), 'w'), indent=1)
    for tag, rows in out.items():
        print(f'\n### {tag}\n| variant | ' + ' | '.join(LOSS_KEYS) + ' |\n|---|' + '---|' * len(LOSS_KEYS))
        for name, e in rows.items():
            print(f'| {name} | ' + ' | '.join(f'{e[k]:.1e}' for k in LOSS_KEYS) + ' |')


if __name__ == '__main__':
    main()
