"""``ExclusiveNCEwithRankingLoss`` (mmaction/models/losses/contrastive_loss.py:72-161):
all-gather the four embeddings, then the fused HIP loss (``clv_infonce_fwd/bwd``)."""
import torch.nn as nn

from .. import ops
from ..builder import LOSSES
from ..utils.dist import get_dist_info
from ..utils.gather_loss import packed_all_gather


@LOSSES.register_module()
class ExclusiveNCEwithRankingLoss(nn.Module):
    def __init__(self, temperature=0.05, use_rank=False, use_rank_ttm=True, use_rank_trtm=True, margin_ttm=5.,
                 margin_trtm=10.):
        super().__init__()
        self.t = temperature
        self.margin_ttm = margin_ttm
        self.margin_trtm = margin_trtm          # constructed but never used by the reference forward (R3)
        self.use_rank = use_rank
        self.use_rank_ttm = use_rank_ttm
        self.use_rank_trtm = use_rank_trtm
        self.fp16_enabled = False
        self.equal_batch = True                  # per-rank batches equal (the trainer's sampler); False -> size exchange

    @property
    def rank(self):
        return get_dist_info()[0]

    @property
    def world_size(self):
        return get_dist_info()[1]

    def forward(self, video_embd=None, text_embd=None, text_mask_embd=None, text_recon_embd=None, **kwargs):
        if any(e is None for e in (video_embd, text_embd, text_mask_embd, text_recon_embd)):
            raise NotImplementedError('the fused loss needs all four embeddings (use_Cmask=True path)')
        v, t, tm, tr = packed_all_gather([video_embd, text_embd, text_mask_embd, text_recon_embd],
                                         equal_sizes=self.equal_batch)
        nce, rank = ops.exclusive_infonce_rank(v, t, tm, tr, self.t, self.margin_ttm)
        return self._pack_losses(nce, rank)

    def _pack_losses(self, nce, rank):
        losses = {'nce_loss': nce}
        if self.use_rank and self.use_rank_ttm:
            losses['rank_t_tm_loss'] = rank
        return losses

    def forward_gathered(self, gathered, slots):
        """The loss on slots (video, text, text_mask, text_recon) of an ALREADY gathered fp32 [G, k, D] tensor: the
        recognizer gathers all its embeddings with one collective and evaluates both directions on it."""
        nce, rank = ops.exclusive_infonce_rank_packed(gathered, slots, self.t, self.margin_ttm)
        return self._pack_losses(nce, rank)

    def forward_gathered_pair(self, gathered, slots_a, slots_b):
        """``forward_gathered`` on two slot quadruples of the same tensor in the same kernel launches (the recognizer's
        video -> text and text -> video evaluations): -> (losses of slots_a, losses of slots_b)."""
        nce_a, rank_a, nce_b, rank_b = ops.exclusive_infonce_rank_pair(gathered, slots_a, slots_b, self.t, self.margin_ttm)
        return self._pack_losses(nce_a, rank_a), self._pack_losses(nce_b, rank_b)


@LOSSES.register_module()
class NormSoftmaxLoss(nn.Module):
    """The retrieval fine-tuning loss (mmaction/models/losses/contrastive_loss.py:26-68): all-gather video / text
    embeddings (``GatherLoss``: the backward keeps the local slice only), normalise, similarity / temperature,
    -mean diag(log_softmax) in both directions.  The arithmetic after the gather is the fused HIP path
    ``clv_normsoftmax_fwd/bwd``."""

    def __init__(self, temperature=0.07, cos_sim=False):
        super().__init__()
        self.t = temperature
        self.use_cos_similarity = cos_sim
        self.fp16_enabled = False
        self.equal_batch = True                  # GatherLoss (:42-43) requires equal per-rank batches

    @property
    def rank(self):
        return get_dist_info()[0]

    @property
    def world_size(self):
        return get_dist_info()[1]

    def forward(self, video_embd=None, text_embd=None, sim_mat=None):
        if sim_mat is not None:                  # :55-56
            return ops.norm_softmax_loss(sim_mat=sim_mat)
        v, t = packed_all_gather([video_embd, text_embd], equal_sizes=self.equal_batch)
        # F.normalize clamps the norm at 1e-12 (:51-52); the cos_sim variant (sim_matrix :10-18) at 1e-8
        return ops.norm_softmax_loss(v, t, temperature=self.t, eps=1e-8 if self.use_cos_similarity else 1e-12)
