"""``CloverFinetune`` — downstream fine-tuning on the pre-trained encoders
(mmaction/models/recognizers/multimodal_transformer_finetune.py:9-215), same constructor kwargs and ``losses`` keys.

The MI355X path covers ``task='retrieval'`` (:83-86 train, :146-148 test): the two uni-modal encoders of the
pre-training step (the HIP Swin + BERT paths) feeding the contrastive projections and ``NormSoftmaxLoss``.  The
``video_qa`` / ``FIB`` tasks need heads (itm_head / qa_head) outside SURVEY §8's scope: the constructor raises
NotImplementedError for them instead of running anything else in their place.
"""
import torch

from .. import ops
from ..builder import RECOGNIZERS, build_backbone, build_head, build_loss
from .base import BaseRecognizer


@RECOGNIZERS.register_module()
class CloverFinetune(BaseRecognizer):
    def __init__(self, mm_backbone, text_backbone=None, freeze_text_backbone=None, loss_type=None, task=None,
                 ssl_head=None, itm_head=None, answer_mask=False, answer_cls=False, qa_head=None,
                 from_scratch=False, text_vocab_size=30522, separate_test=False, **kwargs):
        super().__init__(**kwargs)
        # the reference builds the fusion encoder for every task (:28) although retrieval never runs it; keeping
        # it keeps pre-training checkpoints loadable with strict=True
        self.multimodal_backbone = build_backbone(mm_backbone)
        self.text_backbone = build_backbone(text_backbone)
        self.text_vocab_size = text_vocab_size
        self.from_scratch = from_scratch
        self.separate_test = separate_test
        self.task = task
        if task == 'retrieval':
            self.ssl_head = build_head(ssl_head)
            self.loss_func = build_loss(loss_type)
        elif task in ('video_qa', 'FIB'):
            raise NotImplementedError(f"task={task!r}: only the retrieval fine-tuning task is on the MI355X path")
        else:
            raise NotImplementedError('must have head to do downstream finetuning')       # :45-46
        self.fp16_enabled = False

    def extract_visual_feat(self, imgs):
        return self.backbone(imgs)

    # ---- the engine's split of the step (same contract as CloverPretrain): ``encode`` is everything that touches
    # only this rank's samples (captured as hipGraphs), ``contrastive_losses`` holds the all-gather + the loss
    CLV_ENCODE_KEYS = ('token_ids', 'input_mask')
    EMB_NAMES = ('visual_emb', 'text_emb')

    def _embeddings(self, imgs, token_ids, input_mask, video_cut=None, text_cut=None):
        """Shared by train and test (:61-81 / :128-147): video tokens (mean over the clips of a sample when
        there are several), caption hidden states, then the two projections."""
        if self.training and imgs.is_cuda:
            ops.dropout_seeds_begin(imgs.device)
        imgs = imgs.reshape((-1,) + imgs.shape[2:])                                   # :62
        if self.from_scratch:
            imgs = imgs / 255.0
        B_text = token_ids.shape[0]
        token_ids = token_ids.reshape((-1,) + token_ids.shape[2:])                    # :67-69
        input_mask = input_mask.reshape((-1,) + input_mask.shape[2:])

        def text_side():
            text = self.text_backbone(token_ids, input_mask)['last_hidden_state']     # :78-79
            if text_cut is not None:
                leaf = text.detach().requires_grad_()
                text_cut.append((text, leaf))
                text = leaf
            return self.ssl_head.forward_text(text, input_mask, token_ids)
        # the caption encoder's kernels are tiny (B x L tokens) and independent of the video encoder: run them on
        # a side stream underneath the Swin kernels, as in the pre-training step
        side = self._text_stream(imgs.device) if imgs.is_cuda and getattr(self, 'overlap_text', True) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                text_emb = text_side()
        vis = self.backbone.forward_tokens(imgs, mid_cut=video_cut)                   # channels-last [B,T',h,w,D]
        if video_cut is not None:            # engine graph mode: see CloverPretrain.encode
            cut = (vis, vis.detach().requires_grad_())
            video_cut.append(cut)
            vis = cut[1]
        if B_text != vis.shape[0]:                                                    # :73-75
            vis = vis.reshape((B_text, -1) + vis.shape[1:]).float().mean(dim=1)
        if side is not None:
            main.wait_stream(side)
            text_emb.record_stream(main)
        else:
            text_emb = text_side()
        return self.ssl_head.forward_vision(vis, channels_last=True), text_emb

    def _text_stream(self, device):
        st = getattr(self, '_txt_stream', None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            object.__setattr__(self, '_txt_stream', st)
        return st

    def encode(self, imgs, token_ids=None, input_mask=None, video_cut=None, text_cut=None, **kwargs):
        """-> (emb fp32 [B, 2, D] in EMB_NAMES order, None): the retrieval step has no rank-local loss."""
        v, t = self._embeddings(imgs, token_ids, input_mask, video_cut, text_cut)
        return torch.stack([v, t], dim=1).float(), None

    def contrastive_losses(self, emb, _local_loss=None):
        return {'retrieval_nce_loss': self.loss_func(emb[:, 0], emb[:, 1])}           # :84-86

    def forward_train(self, imgs, label=None, token_ids=None, segment_ids=None, input_mask=None, ans_ids=None,
                      ans_mask=None, **kwargs):
        return self.contrastive_losses(*self.encode(imgs, token_ids=token_ids, input_mask=input_mask))

    def forward_test(self, imgs, token_ids=None, segment_ids=None, input_mask=None, ans_ids=None, ans_mask=None,
                     **kwargs):
        if not self.separate_test:
            raise NotImplementedError('not implement the finetune test method')       # :203-204
        return self._embeddings(imgs, token_ids, input_mask)                          # :146-148

    def forward_gradcam(self, imgs, token_ids=None, segment_ids=None, input_mask=None):
        return self.forward_test(imgs, token_ids, segment_ids, input_mask)
