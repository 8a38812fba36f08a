"""``CloverPretrain`` — the video-text pre-training step graph of
mmaction/models/recognizers/multimodal_transformer_pretrain.py:11-173, same constructor
kwargs, same ``losses`` keys.

MI355X-first scheduling of the SAME arithmetic (every encoder is sample-independent —
LayerNorm only — so batching passes along dim 0 changes no value):
  * clean + masked video passes (:91,114)  -> one 2B-clip Swin pass (``forward_both``)
  * un-masked + masked caption passes (:99,110) -> one 2B BERT pass
  * v_fusion + t_fusion (:117,119)          -> one 2B fusion pass
  * MLM head only on the t_fusion half; all B*L rows go through the decoder and the fused
    focal kernel skips label == -100 rows (:134-139) — no data-dependent shapes, no host sync.
"""
import os

import torch

from .. import ops

from ..builder import RECOGNIZERS, build_backbone, build_head, build_loss
from .base import BaseRecognizer


@RECOGNIZERS.register_module()
class CloverPretrain(BaseRecognizer):
    def __init__(self, mm_backbone, text_backbone=None, freeze_text_backbone=None, freeze_dvae_backbone=None,
                 loss_type=None, ssl_loss=None, ssl_head=None, mlm_head=None, mlm_loss=None, mlm_ssl_head=None,
                 symmetry_rank=False, separate_test=False, from_scratch=False, use_Cmask=True,
                 text_vocab_size=30522, **kwargs):
        super().__init__(**kwargs)
        self.multimodal_backbone = build_backbone(mm_backbone)
        self.text_backbone = build_backbone(text_backbone)
        self.text_vocab_size = text_vocab_size
        self.loss_func = build_loss(loss_type) if loss_type is not None else None
        self.use_Cmask = use_Cmask
        self.mlm_head = build_head(mlm_head) if mlm_head is not None else None
        if mlm_ssl_head is not None:
            self.mlm_ssl_V_head = build_head(mlm_ssl_head['V']) if mlm_ssl_head.get('V') else None
            self.mlm_ssl_T_head = build_head(mlm_ssl_head['T']) if mlm_ssl_head.get('T') else None
        else:
            self.mlm_ssl_V_head = None
            self.mlm_ssl_T_head = None
        self.mlm_loss_func = build_loss(mlm_loss) if mlm_loss is not None else None
        self.symmetry_rank = symmetry_rank
        self.from_scratch = from_scratch
        self.separate_test = separate_test
        if ssl_head is not None:
            self.ssl_head_name = ssl_head['type']
            self.ssl_head = build_head(ssl_head)
            self.ssl_loss = build_loss(ssl_loss)
        self.fp16_enabled = False
        if freeze_dvae_backbone is not None:
            self._freeze(freeze_stage=freeze_dvae_backbone, freeze_except=[])
        if freeze_text_backbone is not None:
            self._freeze(freeze_stage=freeze_text_backbone, freeze_except=[])

    def extract_visual_feat(self, imgs, mask=None):
        return self.backbone(imgs, mask)

    EMB_NAMES = ('visual_emb', 'text_emb', 'mask_word_emb', 'mask_visual_recon_emb', 'mask_visual_emb',
                 'mask_word_recon_emb')

    def encode(self, imgs, token_ids=None, input_mask=None, mlm_label=None, v_token_mask=None, video_cut=None,
               text_cut=None, **kwargs):
        """Everything of the step that touches only THIS rank's samples: the three encoders, the heads
        and the MLM loss.  Returns (emb fp32 [B, 6, D] in EMB_NAMES order, mlm_loss).  No collective
        and no data-dependent shape inside — the engine captures it (and its backward) as hipGraphs."""
        # Ablation switches of forward_train (:129-169): mlm_head=None (no MLM loss), mlm_ssl_head=None (no V / T
        # reconstruction heads: the first ssl_loss call is skipped, :147), symmetry_rank=False (no second ssl_loss call, :155),
        # mlm_loss=None (CrossEntropyLoss of loss_type over the rows, :141-142).  What the reference itself cannot run is
        # refused with its reason: use_Cmask=False hands None to cos_norm (contrastive_loss.py:114-115), symmetry_rank
        # without a T head calls None (:157), and mlm_label / v_token_mask are read unconditionally (:97, :114).
        if not hasattr(self, 'ssl_head'):
            raise NotImplementedError('CloverPretrain without ssl_head: forward_train reads visual_emb / text_emb of the '
                                      'ssl_head branch unconditionally (:102, :151)')
        if not self.use_Cmask and (self.mlm_ssl_V_head is not None or self.symmetry_rank):
            raise ValueError('use_Cmask=False passes text_mask_embd=None into ExclusiveNCEwithRankingLoss, whose cos_norm '
                             '(contrastive_loss.py:114) fails on None: not a runnable configuration of the reference')
        if self.symmetry_rank and self.mlm_ssl_T_head is None:
            raise ValueError('symmetry_rank=True needs mlm_ssl_head["T"] (multimodal_transformer_pretrain.py:157)')
        if mlm_label is None or v_token_mask is None:
            raise ValueError('forward_train needs mlm_label and v_token_mask (multimodal_transformer_pretrain.py:97, :114)')
        if self.training and imgs.is_cuda:
            ops.dropout_seeds_begin(imgs.device)        # one RNG-counter kernel for all dropout sites of the step
        imgs = imgs.reshape((-1,) + imgs.shape[2:])                                   # :81
        if self.from_scratch:
            imgs = imgs / 255.0
        token_ids = token_ids.reshape((-1,) + token_ids.shape[2:])                    # :85
        text_input_mask = input_mask.reshape((-1,) + input_mask.shape[2:])
        mlm_label = mlm_label.reshape((-1,) + mlm_label.shape[2:])
        B = imgs.shape[0]

        # ---- text encoder (:97-101, :110-111) on a SIDE stream: it is independent of the video encoder until the
        # projection heads, and its ~250 kernels are tiny (16 x 32 tokens: 36..144 workgroups each), so they
        # run in the shadow of the Swin kernels instead of serially after them.  Autograd replays each backward
        # on its forward's stream, so the two backward passes overlap the same way; a hipGraph capture records
        # the fork / join as graph edges.
        # Batch layout of the doubled passes (no cat / slice copies anywhere downstream):
        #   video  y        = [clean clips (:91)      ; masked clips (:114)]
        #   text   text_out = [masked caption (:110)  ; un-masked caption (:99)]
        # so that row block 0 of the fusion pass is t_fusion = (clean video, masked text) (:119) and row block 1 is
        # v_fusion = (masked video, clean text) (:117) with both inputs used exactly as the encoders produced them.
        # which block of a doubled projection-head pass the reference itself runs (BatchNorm variants keep the running
        # statistics of the others frozen): the masked caption only with mlm_ssl_V_head (:147-150), the masked clip only with
        # symmetry_rank (:155-159)
        text_live = (1,) + ((0,) if self.mlm_ssl_V_head is not None else ())
        vis_live = (0,) + ((1,) if self.symmetry_rank else ())

        def text_inputs():
            input_ssl_ids = torch.where(mlm_label == -100, token_ids, mlm_label)
            return torch.cat([token_ids, input_ssl_ids], 0), torch.cat([text_input_mask, text_input_mask], 0)
        side = self._text_stream(imgs.device) if imgs.is_cuda and getattr(self, 'overlap_text', True) else None
        if side is not None:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                text_ids2, text_mask2 = text_inputs()        # the text tower's own input glue: off the video encoder's stream
                text_out = self._cut(self.text_backbone(text_ids2, text_mask2)['last_hidden_state'], text_cut)
                txt_emb_both = self.ssl_head.forward_text(text_out, passes=2, order=(1, 0), live=text_live)   # :150 / :102, also text-only
                fusion_prep = None
                if os.environ.get('CLOVER_HEADS_SIDE', '1') == '1' and hasattr(self.multimodal_backbone, 'prepare'):
                    # the video-independent part of the fusion encoder's input (text + type embeddings, position table,
                    # key mask) here, behind the text encoder, not between the video encoder and the first fusion layer
                    mb = self.multimodal_backbone
                    fusion_prep = mb.prepare(text_out, text_mask2, 2 * B, (imgs.shape[2] + 1) // 2, mb.spacial_tokens)

        # ---- video encoder: clean (:91) + masked (:114) pass as one 2B-clip pass, channels-last [2B,T',h,w,Cf]
        vis_both = self.backbone.forward_both(imgs, v_token_mask, mid_cut=video_cut)
        if video_cut is not None:
            # engine graph mode (data parallel): the autograd graph is cut at the encoders' outputs, so that each
            # encoder's backward is a graph of its own and gradient buckets can leave between them.
            # A cut list receives [(output, detached leaf standing in for it downstream), ...]
            vis_both = self._cut(vis_both, video_cut)
        _, T, h, w, D = vis_both.shape

        # ---- text encoder: masked caption (:110-111) + un-masked caption (:97-101)
        if side is not None:
            main.wait_stream(side)
            text_out.record_stream(main)
            txt_emb_both.record_stream(main)
            text_mask2.record_stream(main)
        else:
            text_ids2, text_mask2 = text_inputs()
            text_out = self._cut(self.text_backbone(text_ids2, text_mask2)['last_hidden_state'], text_cut)
            txt_emb_both = self.ssl_head.forward_text(text_out, passes=2, order=(1, 0), live=text_live)   # (reference order: un-masked :102, masked :150)
            fusion_prep = None

        # ---- contrastive projections (:102, :150, :159); unbind of a [2, B, ..] view: its backward is one stack.
        # The vision projection head (pool + 2 Linear + 2 LayerNorm + GELU: ~10 launch-bound kernels) feeds only the loss:
        # it runs on the side stream under the fusion encoder (and, autograd replaying it there, its backward under the
        # fusion encoder's backward) instead of between the video encoder and the fusion encoder.
        heads_side = side is not None and os.environ.get('CLOVER_HEADS_SIDE', '1') == '1'
        if heads_side:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                vis_both.record_stream(side)
                vis_emb_both = self.ssl_head.forward_vision(vis_both, channels_last=True, passes=2, live=vis_live)
        else:
            vis_emb_both = self.ssl_head.forward_vision(vis_both, channels_last=True, passes=2, live=vis_live)
        mask_word_emb, text_emb = txt_emb_both.view(2, B, -1).unbind(0)

        # ---- fusion: block 0 = t_fusion (clean video, masked text) (:119); block 1 = v_fusion (masked video,
        # clean text) (:117)
        if side is not None and fusion_prep is not None:
            for v_ in fusion_prep.values():
                if torch.is_tensor(v_):
                    v_.record_stream(main)
        # flush_point(aux): the fusion encoder's backward ends at its visual input — the weight gradients of the heads and of
        # the fusion encoder, complete by then, leave on an auxiliary stream under the video tower's backward
        fusion = self.multimodal_backbone(visual_token=ops.flush_point(vis_both.reshape(2 * B, T, h * w, D), aux=True),
                                          text_input_mask=text_mask2, text_input_embeds=text_out,
                                          prepared=fusion_prep if side is not None else None)
        # row block 0 = t_fusion, block 1 = v_fusion: the caption tokens of block 0 feed the MLM decoder (:129), the caption
        # CLS rows of both blocks the reconstruction heads (:148-149, :156-157) — one autograd node (ops.fusion_text_outputs)
        h_all = fusion['last_hidden_state']
        t_last_hidden_state, cls_rows = ops.fusion_text_outputs(h_all, h_all.shape[1] - fusion['t_last_hidden_state'].shape[1])

        # ---- the two reconstruction heads (:148-149, :156-157; a dozen launch-bound kernels) on the side stream, under
        # the MLM decoder GEMM + focal loss
        def recon_heads():
            # a switched-off head leaves its slot of the packed embeddings zero (no gradient): contrastive_losses skips
            # the loss call that would read it
            mvr = self.mlm_ssl_V_head(cls_rows[1]) if self.mlm_ssl_V_head is not None else None             # :148-149
            mwr = self.mlm_ssl_T_head(cls_rows[0]) if self.symmetry_rank else None                          # :156-157
            return mvr, mwr
        if heads_side:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                cls_rows.record_stream(side)
                mask_visual_recon_emb, mask_word_recon_emb = recon_heads()

        # ---- MLM (:129-143): all B*L rows through the decoder, the fused focal kernel skips label == -100
        mlm_loss = None
        if self.mlm_head is not None:
            score = self.mlm_head(t_last_hidden_state)
            fn = self.mlm_loss_func if self.mlm_loss_func is not None else self.loss_func
            mlm_loss = fn(score.reshape(-1, self.text_vocab_size), mlm_label.reshape(-1))

        if heads_side:
            main.wait_stream(side)
            for t in (vis_emb_both, mask_visual_recon_emb, mask_word_recon_emb):
                if t is not None:
                    t.record_stream(main)
        else:
            mask_visual_recon_emb, mask_word_recon_emb = recon_heads()
        visual_emb, mask_visual_emb = vis_emb_both.view(2, B, -1).unbind(0)
        if mask_visual_recon_emb is None:
            mask_visual_recon_emb = torch.zeros_like(visual_emb)
        if mask_word_recon_emb is None:
            mask_word_recon_emb = torch.zeros_like(visual_emb)
        emb = torch.stack([visual_emb, text_emb, mask_word_emb, mask_visual_recon_emb, mask_visual_emb,
                           mask_word_recon_emb], dim=1).float()
        return emb, mlm_loss

    @staticmethod
    def _cut(t, cuts):
        if cuts is None:
            return t
        leaf = t.detach().requires_grad_()
        cuts.append((t, leaf))
        return leaf

    def _text_stream(self, device):
        st = getattr(self, '_txt_stream', None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            object.__setattr__(self, '_txt_stream', st)
        return st

    def contrastive_losses(self, emb, mlm_loss, gathered=None):
        """The cross-rank part of the step (:147-169): ONE all-gather of the six embeddings (backward = the local
        slice, gather_loss.py:64-72), then the two exclusive-InfoNCE / ranking evaluations read their four slots
        of the gathered [G, 6, D] tensor in place.  emb [B, 6, D] in EMB_NAMES order."""
        from ..utils.gather_loss import gather_rows
        # gathered: the [G, 6, D] fp32 tensor already in hand (the engine's loss graph gathers outside the capture)
        g = (gathered if gathered is not None
             else gather_rows(emb.float(), equal_sizes=self.ssl_loss.equal_batch).contiguous())
        V, T, MW, MVR, MV, MWR = range(6)                      # EMB_NAMES order
        losses = dict(mlm_loss=mlm_loss) if mlm_loss is not None else {}
        pair = (self.mlm_ssl_V_head is not None and self.symmetry_rank and g.is_cuda
                and os.environ.get('CLOVER_LOSS_PAIR', '1') == '1')
        if pair:                                               # both evaluations in the same kernel launches
            l1, l2 = self.ssl_loss.forward_gathered_pair(g, (V, T, MW, MVR), (T, V, MV, MWR))              # :151, :161
            losses.update(l1)
        elif self.mlm_ssl_V_head is not None:                                                              # :147
            losses.update(self.ssl_loss.forward_gathered(g, (V, T, MW, MVR)))                              # :151
        if self.symmetry_rank:                                                                             # :155
            if not pair:
                l2 = self.ssl_loss.forward_gathered(g, (T, V, MV, MWR))                                    # :161
            l2['v_nce_loss'] = l2.pop('nce_loss')
            if self.ssl_loss.use_rank:
                l2['rank_v_vm_loss'] = l2.pop('rank_t_tm_loss')
            losses.update(l2)
        return losses

    def forward_train(self, imgs, label, token_ids=None, segment_ids=None, input_mask=None, mlm_label=None,
                      dvae_imgs=None, v_token_mask=None, hog_features=None, img_metas=None, **kwargs):
        emb, mlm_loss = self.encode(imgs, token_ids=token_ids, input_mask=input_mask, mlm_label=mlm_label,
                                    v_token_mask=v_token_mask)
        return self.contrastive_losses(emb, mlm_loss)

    def forward_test(self, imgs, token_ids=None, segment_ids=None, input_mask=None, **kwargs):
        """``separate_test`` inference (:194-218): one Swin pass + one BERT pass -> (visual_emb, text_emb)."""
        if not self.separate_test:
            raise NotImplementedError('only separate_test=True inference exists in the reference (R4)')
        imgs = imgs.reshape((-1,) + imgs.shape[2:])
        if self.from_scratch:
            imgs = imgs / 255.0
        vis = self.backbone.forward_tokens(imgs)
        B_text = token_ids.shape[0]
        if B_text != vis.shape[0]:
            vis = vis.reshape((B_text, -1) + vis.shape[1:]).float().mean(dim=1)
        token_ids = token_ids.reshape((-1,) + token_ids.shape[2:])
        input_mask = input_mask.reshape((-1,) + input_mask.shape[2:])
        text = self.text_backbone(token_ids, input_mask)['last_hidden_state']
        return self.ssl_head.forward_vision(vis, channels_last=True), self.ssl_head.forward_text(text, input_mask, token_ids)

    def forward_gradcam(self, imgs, **kwargs):
        raise NotImplementedError
