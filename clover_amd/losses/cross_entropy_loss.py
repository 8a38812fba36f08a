"""``CrossEntropyLoss`` (mmaction/models/losses/cross_entropy_loss.py:10-83).  Built by every
Clover config as ``loss_type`` but only evaluated when ``mlm_loss`` is None
(multimodal_transformer_pretrain.py:141-142); it is the gamma = 0 case of the fused focal
kernel (ignore_index -100, mean over the kept rows)."""
import torch.nn as nn

from .. import ops
from ..builder import LOSSES


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, loss_weight=1.0, class_weight=None):
        super().__init__()
        if class_weight is not None:
            raise NotImplementedError('class_weight is not used by the pre-training path')
        self.loss_weight = loss_weight
        self.class_weight = None

    def forward(self, cls_score, label, **kwargs):
        return self.loss_weight * ops.focal_ce_masked(cls_score.reshape(-1, cls_score.shape[-1]),
                                                      label.reshape(-1), 0.0)
