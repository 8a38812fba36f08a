"""ORACLE (test infrastructure, never on the product path): the reference's loss scaler restated.

Follows mmaction/core/hooks/fp16_utils.py:285-389 (``LossScaler``: ``__init__`` :314-327, ``update_scale`` :351-362,
``state_dict`` :364-373) as used by ``Fp16OptimizerHook.after_train_iter`` (mmcv_Fp16OptimizerHook.py:96-149): the loss is
multiplied by ``cur_scale``; an iteration whose gradients overflow is skipped and divides the scale by ``scale_factor``
(floor 1); ``scale_window`` iterations after the last overflow the scale is multiplied by it.

Pinned: tests/test_oracle_golden.py::test_loss_scaler_matches_reference_trajectories compares it step by step with
trajectories of the reference class itself (tests/golden/g_scaler.npz, written by tests/golden/make_goldens.py scaler).
The product's scaler is the second half of ``OptimState`` in clover_amd/csrc/optim.hip (device resident).
"""


class LossScaler:
    def __init__(self, init_scale=2.0 ** 32, mode='dynamic', scale_factor=2.0, scale_window=1000):
        if mode not in ('dynamic', 'static'):
            raise AssertionError('mode can only be dynamic or static')
        self.cur_scale = init_scale
        self.cur_iter = 0
        self.last_overflow_iter = -1
        self.mode = mode
        self.scale_factor = scale_factor
        self.scale_window = scale_window

    @property
    def loss_scale(self):
        return self.cur_scale

    def update_scale(self, overflow):
        if self.mode != 'dynamic':
            return                                   # (a static scaler does not even count iterations)
        if overflow:
            self.cur_scale = max(self.cur_scale / self.scale_factor, 1)
            self.last_overflow_iter = self.cur_iter
        elif (self.cur_iter - self.last_overflow_iter) % self.scale_window == 0:
            self.cur_scale *= self.scale_factor
        self.cur_iter += 1

    def state_dict(self):
        return dict(cur_scale=self.cur_scale, cur_iter=self.cur_iter, mode=self.mode,
                    last_overflow_iter=self.last_overflow_iter, scale_factor=self.scale_factor,
                    scale_window=self.scale_window)
