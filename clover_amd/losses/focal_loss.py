"""``SoftmaxFocalLossMultiClass`` (mmaction/models/losses/focal_loss.py:50-72), fused:
one HIP pass over the [rows, V] logits does log-softmax + label gather + focal weighting.
Rows whose target is -100 are skipped, so the caller may pass ALL B*L rows instead of
gathering the masked ones first (multimodal_transformer_pretrain.py:137-139) — same mean,
no data-dependent shape, no host sync."""
import torch.nn as nn

from .. import ops
from ..builder import LOSSES


@LOSSES.register_module()
class SoftmaxFocalLossMultiClass(nn.Module):
    def __init__(self, gamma=2.0, reduction='mean'):
        super().__init__()
        if reduction != 'mean':
            raise NotImplementedError("reduction='sum' is not used by the pre-training path")
        self.gamma = gamma
        self.reduction = reduction
        self.fp16_enabled = False

    def forward(self, input, target):
        return ops.focal_ce_masked(input.reshape(-1, input.shape[-1]), target.reshape(-1), self.gamma)
