from .dist import get_dist_info  # noqa: F401
from .gather_loss import GatherLoss, VariedShapeGatherLoss, packed_all_gather  # noqa: F401
