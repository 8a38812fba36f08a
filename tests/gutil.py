"""Helpers shared by the parity tests: fixture loading + packed-tensor comparison."""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
MAXSUB = 4096


def load(name):
    return np.load(os.path.join(GOLD, name))


def manifest():
    with open(os.path.join(GOLD, 'manifest_tiny.json')) as f:
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
