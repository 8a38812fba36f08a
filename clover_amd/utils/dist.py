"""mmcv.runner.get_dist_info equivalent (used by contrastive_loss.py:35,77 in the reference)."""
import torch.distributed as dist


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collectives_active():
    """True when the cross-rank collectives must really be issued: W > 1, or W == 1 with
    CLOVER_FORCE_COLLECTIVES=1 (lets a 1-GPU box exercise the RCCL code path end to end)."""
    import os
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get('CLOVER_FORCE_COLLECTIVES') == '1'
