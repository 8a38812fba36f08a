"""Building blocks shared by the registered modules: nn.Linear / nn.LayerNorm subclasses
(so mmcv-style param-group rules that test ``isinstance(m, nn.LayerNorm)`` keep working and
state_dict keys stay ``weight``/``bias``) whose forward runs the bf16 HIP kernels of ``clover_amd.ops`` (no library GEMM:
a shape the kernels do not cover goes through ``ops._library_gemm``, which counts and announces it).

Precision policy (maps the reference's fp16 policy, fp16_utils.py:215-259, to bf16):
parameters are fp32 masters; GEMM operands are bf16 (fp32 accumulate in MFMA); LayerNorm
parameters/statistics, softmax, and every loss are fp32.
"""
import torch
import torch.nn as nn

from . import ops

BF16 = ops.BF16          # the 16-bit storage type: bf16, or fp16 with CLOVER_HALF=f16 (the name is historical)


def to_bf16(t):
    """Activation storage type of the path: bf16 (fp32 in parity mode, clover_amd/parity.py)."""
    if ops.parity.enabled():
        return ops.parity.f32(t)
    return t if t.dtype == BF16 else t.to(BF16)


class Linear(nn.Linear):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # the engine keeps a bf16 W^T where the input gradient's kernel takes one (ops.linear_dgrad)
        self.weight._clv_want_t = ops.wants_transposed(self.out_features, self.in_features)

    def forward(self, x):
        return ops.linear(x, self.weight, self.bias)


class LinearFP32(nn.Linear):
    """fp32 GEMM for the [B, D] projection heads: the contrastive logits are cos/0.05, so a bf16
    rounding of an embedding (2^-9 relative) would already move the loss by ~1e-2; these GEMMs are
    O(B*D^2) and cost nothing (the reference forces fp32 here too, contrastive_loss.py:102)."""

    def forward(self, x):
        return ops.linear_f32(x, self.weight, self.bias)             # exact-f32 MFMA kernel (raises on a CPU tensor)


class LayerNorm(nn.LayerNorm):
    def forward(self, x, residual=None, return_sum=False, x_scale=None, x_dropout_p=0.0, fork=False):
        return ops.layer_norm(x, self.weight, self.bias, self.eps, residual=residual, return_sum=return_sum,
                              x_scale=x_scale, x_dropout_p=x_dropout_p, fork=fork)


class BatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters / buffers / state_dict keys) whose forward is the HIP kernel; [B, D] rows only
    (the projection heads' use, ssl_head.py:52,56,60,175-186)."""

    _frozen = 0          # > 0 inside BatchNorm1d.frozen_stats(): batch statistics are used, the running ones are not touched

    @classmethod
    def frozen_stats(cls):
        """Context: training-mode calls normalise with their batch statistics but leave running_mean / running_var /
        num_batches_tracked alone — for a pass the reference does not run (the recognizer's doubled clean + masked batch
        under an ablation switch): its outputs are computed, its inputs never reach the layer's state."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            cls._frozen += 1
            try:
                yield
            finally:
                cls._frozen -= 1
        return ctx()

    def forward(self, x):
        if self.momentum is None or not self.affine or not self.track_running_stats:
            raise NotImplementedError('BatchNorm1d(momentum=None / affine=False / track_running_stats=False)')
        frozen = BatchNorm1d._frozen > 0
        if self.training and not frozen:
            self.num_batches_tracked.add_(1)
        # momentum 0: running = 1 * running + 0 * batch — exactly unchanged
        return ops.batch_norm1d(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                0.0 if frozen else self.momentum, self.eps)


class GELU(nn.Module):
    def forward(self, x):
        return ops.gelu(x)


class DropPath(nn.Module):
    """Per-sample stochastic depth (timm DropPath; swin_transformer_3d.py:441,498,503).

    ``scale(x)`` returns the per-sample factor [B] (mask / keep_prob, fp32) or None when inactive, so that a
    consumer can fold it into its own kernel (the LayerNorm that follows: ``x_scale=``).  A parent module may
    preset the factors of all its DropPaths with one RNG call per step (``preset``)."""

    def __init__(self, drop_prob=0.):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self._preset = None

    def preset(self, scale):
        self._preset = scale

    def scale(self, x):
        if self.drop_prob == 0. or not self.training:
            return None
        if self._preset is not None and self._preset.shape[0] == x.shape[0]:
            return self._preset
        keep = 1.0 - self.drop_prob
        return torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep) / keep

    @staticmethod
    def apply_scale(x, scale):
        return x if scale is None else x * scale.view((x.shape[0],) + (1,) * (x.dim() - 1)).to(x.dtype)

    def forward(self, x):
        return self.apply_scale(x, self.scale(x))

    def extra_repr(self):
        return f'drop_prob={self.drop_prob}'


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)
