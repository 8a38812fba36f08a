from .contrastive_loss import ExclusiveNCEwithRankingLoss, NormSoftmaxLoss
from .cross_entropy_loss import CrossEntropyLoss
from .focal_loss import SoftmaxFocalLossMultiClass

__all__ = ['ExclusiveNCEwithRankingLoss', 'NormSoftmaxLoss', 'CrossEntropyLoss', 'SoftmaxFocalLossMultiClass']
