# Retrieval fine-tuning (VideoSwin-T + BERT-base, NormSoftmaxLoss) in the reference's config format
# (configs/exp_local/finetune_msrvtt_retrieval.py:7-74, with the Swin-T widths of BASELINE config 2); the MSRVTT
# loaders are replaced by a synthetic batch stream of the same layout.  `load_from` takes a pre-training checkpoint
# written by tools/train.py (the fusion encoder's weights are loaded and left untouched, as in the reference).
_base_ = ['_base_default_runtime.py']
videos_per_gpu = 16
num_frames = 8
base_lr = 1.2e-5 / 128
weight_decay = 0.01
import bench as _bench                                   # noqa: E402  (repo root is on sys.path under tools/train.py)
_pre = _bench.model_cfg('T', num_frames)
model = dict(type='CloverFinetune', freeze_stage=None, separate_test=True, backbone=_pre['backbone'],
             freeze_text_backbone=None, text_vocab_size=30522, mm_backbone=_pre['mm_backbone'],
             text_backbone=_pre['text_backbone'], cls_head=None, task='retrieval', ssl_head=_pre['ssl_head'],
             itm_head=None, loss_type=dict(type='NormSoftmaxLoss', cos_sim=True, temperature=0.05),
             train_cfg=dict(aux_info=['token_ids', 'segment_ids', 'input_mask']),
             test_cfg=dict(feature_extraction=False))
del _pre
data = dict(videos_per_gpu=videos_per_gpu, synthetic=[dict(length=20, frames=num_frames, tokens=32)],
            synthetic_test=dict(pairs=200, frames=num_frames, tokens=32))          # tools/test.py
evaluation = dict(interval=1, metrics=['recall_for_video_text_retrieval'], gpu_collect=True)
optimizer = dict(type='AdamW', base_lr=base_lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=weight_decay,
                 paramwise_cfg=dict(norm_decay_mult=0.0, bias_decay_mult=0.0,
                                    custom_keys={'absolute_pos_embed': dict(decay_mult=0.),
                                                 'relative_position_bias_table': dict(decay_mult=0.)}))
optimizer_config = dict(grad_clip=dict(max_norm=5))
lr_config = dict(policy='CosineAnnealing', min_lr_ratio=1e-3, by_epoch=False, warmup='linear', warmup_iters=1,
                 warmup_ratio=0.001, warmup_by_epoch=True)
total_epochs = 2
workflow = [('train', 1)]
