"""Helpers shared by the parity tests: fixture loading + packed-tensor comparison."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MAXSUB = 4096


def load(name):
    return np.load(os.path.join(GOLD, name))


def manifest(which='tiny'):
    with open(os.path.join(GOLD, f'manifest_{which}.json')) as f:
        return json.load(f)


def unused_params():
    with open(os.path.join(GOLD, 'unused_params_tiny.json')) as f:
        return json.load(f)


def packed(t):
    a = t.detach().cpu().double().numpy().reshape(-1)
    stride = max(1, a.size // MAXSUB)
    return a[::stride][:MAXSUB], np.array([a.sum(), np.sqrt((a * a).sum()), a.size])


def assert_packed(g, name, t, rtol=1e-4, atol=1e-5):
    """Compare tensor t against a fixture entry written by make_goldens.pack()."""
    sub, stats = packed(t)
    gsub, gstats = g[name + '.sub'].astype(np.float64), g[name + '.stats']
    assert gstats[2] == stats[2], f'{name}: numel {stats[2]} vs golden {gstats[2]}'
    scale = max(np.abs(gsub).max(), 1e-30)
    err = np.abs(sub - gsub).max()
    assert err <= atol + rtol * scale, f'{name}: max|d|={err:.3e} (scale {scale:.3e})'
    l2 = max(gstats[1], 1e-30)
    assert abs(stats[1] - gstats[1]) <= (atol + rtol * l2) * 10, f'{name}: l2 {stats[1]} vs {gstats[1]}'


def assert_close(a, b, rtol=1e-4, atol=1e-5, name=''):
    a = np.asarray(a.detach().cpu().double() if isinstance(a, torch.Tensor) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, f'{name}: shape {a.shape} vs {b.shape}'
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= atol + rtol * scale, f'{name}: max|d|={err:.3e} (scale {scale:.3e})'


_FULL = {}


def full_size_oracle(variant, frames, B=2):
    """The fp32 oracle's step (losses + every parameter gradient) at a BASELINE configuration's full shapes — seeded
    random init (4321), the synthetic batch of seed 77 — computed once per (variant, frames) and shared by the bf16 and
    fp8 step tests.  -> (cfg, state_dict on CPU, batch on CPU, reference log_vars, {name: reference gradient})."""
    key = (variant, frames, B)
    if key not in _FULL:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        import clover_amd
        from oracle import model as om
        torch.manual_seed(4321)
        cfg = bench.model_cfg(variant, frames)
        m = clover_amd.build_model(cfg).eval()
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        del m
        P = {k: v.detach().float().clone().requires_grad_(v.is_floating_point())
             for k, v in sd.items() if 'relative_position_index' not in k}
        batch = bench.synthetic_batch(B, frames, 32, seed=77)
        torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
        loss, lv = om.parse_losses(om.forward_train(P, batch, bench.oracle_cfg(cfg), gather=False))
        loss.backward()
        grads = {k: v.grad.detach().clone() for k, v in P.items() if v.grad is not None}
        _FULL[key] = (cfg, sd, batch, {k: float(v) for k, v in lv.items()}, grads)
    return _FULL[key]
