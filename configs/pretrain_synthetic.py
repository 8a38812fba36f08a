# BASELINE config 2 (VideoSwin-T + BERT-base + 3-layer fusion, all five losses) in the reference's config format
# (configs/exp_local/pretrain_webvid_cc3m.py:22-141); the two training sets are synthetic (an 8-frame video stream and a
# 1-frame image stream, as WebVid + CC3M; the engine keeps one set of hipGraphs per batch geometry), everything else keeps the reference's keys and values.
_base_ = ['_base_default_runtime.py']
videos_per_gpu = 8
base_lr = 5e-5 / 1024
weight_decay = 0.005
fp16 = dict(loss_scale='dynamic')                       # as the reference (pretrain_webvid_cc3m.py:21): the engine's device-resident scaler
import bench as _bench                                   # noqa: E402  (repo root is on sys.path under tools/train.py)
model = _bench.model_cfg('T', 8)
data = dict(videos_per_gpu=videos_per_gpu,
            # lengths: the reference's interleave (clover_runner.py:76-93) needs long <= 1.5 * short, else its restarted
            # iterator runs dry mid-epoch (StopIteration there and here)
            synthetic=[dict(length=16, frames=8, tokens=32), dict(length=12, frames=1, tokens=32)])
optimizer = dict(type='AdamW', base_lr=base_lr, betas=(0.9, 0.98), eps=1e-8, weight_decay=weight_decay,
                 paramwise_cfg=dict(norm_decay_mult=0.0, bias_decay_mult=0.0,
                                    custom_keys={'absolute_pos_embed': dict(decay_mult=0.),
                                                 'relative_position_bias_table': dict(decay_mult=0.)}))
optimizer_config = dict(grad_clip=dict(max_norm=15))
lr_config = dict(policy='CosineAnnealing', min_lr_ratio=1e-3, by_epoch=False, warmup='linear', warmup_iters=4,
                 warmup_ratio=0.001, warmup_by_epoch=True)
total_epochs = 2
