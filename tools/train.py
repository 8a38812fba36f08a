#!/usr/bin/env python
"""tools/train.py-style driver (reference: tools/train.py:26-88 arguments, :259-340 main, :101-256 train_model).

    python tools/train.py configs/pretrain_synthetic.py --launcher none --cfg-options total_epochs=1
    python -m torch.distributed.run --nproc-per-node 8 tools/train.py configs/pretrain_synthetic.py --launcher pytorch

Same config format (`_base_`, `model = dict(type='CloverPretrain', ...)`, `optimizer` with `base_lr`, `lr_config`,
`total_epochs`, `workflow`, `checkpoint_config`, `log_config`), same CLI names.  The dataset side (decoding,
tokenisation, augmentation pipelines) is outside this project's scope: `data.synthetic` describes loaders of
synthetic batches with the real batch layout instead."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                            # noqa: E402
import torch.distributed as dist                        # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='Train a recognizer')
    p.add_argument('config', help='train config file path')
    p.add_argument('--work_dir', help='the dir to save logs and models')
    p.add_argument('--resume-from', help='the checkpoint file to resume from')
    p.add_argument('--load-from', help='the checkpoint file to load from')
    p.add_argument('--seed', type=int, default=None, help='random seed')
    p.add_argument('--cfg-options', nargs='+', default=[], help='a.b=c overrides merged into the config')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='pytorch', help='job launcher')
    return p.parse_args()


class SyntheticLoader:
    """`length` batches of synthetic pairs with the layout of the reference's collated batch."""

    def __init__(self, length, batch, frames, tokens, seed, device):
        import bench
        self.length = length
        self.batches = [{k: v.to(device) for k, v in bench.synthetic_batch(batch, frames, tokens, seed + i).items()}
                        for i in range(min(length, 4))]

    def __len__(self):
        return self.length

    def __iter__(self):
        for i in range(self.length):
            yield self.batches[i % len(self.batches)]


def main():
    args = parse_args()
    from clover_amd.runner import (CheckpointHook, CloverRunner, Config, LogHook, LrUpdaterHook, parse_cfg_options,
                                   scaled_lr)
    import clover_amd
    from clover_amd.engine import CloverEngine

    cfg = Config.fromfile(args.config)
    cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    if args.work_dir is not None:
        cfg.work_dir = args.work_dir
    elif cfg.get('work_dir') is None:
        cfg.work_dir = os.path.join('./work_dirs', os.path.splitext(os.path.basename(args.config))[0])
    if not torch.cuda.is_available():
        raise SystemExit('tools/train.py needs an MI355X (no CPU fallback)')
    if args.launcher == 'none':
        rank, world = 0, 1
        torch.cuda.set_device(0)
    else:
        rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', rank=rank, world_size=world,
                                device_id=torch.device('cuda', torch.cuda.current_device()))
    dev = torch.device('cuda', torch.cuda.current_device())
    if args.seed is not None:
        torch.manual_seed(args.seed)
    else:
        torch.manual_seed(0)                             # identical init on every rank (== DDP broadcast)

    model = clover_amd.build_model(cfg.model.copy() if hasattr(cfg.model, 'copy') else dict(cfg.model)).to(dev)
    model.train()
    lr = scaled_lr(cfg, world)                           # tools/train.py:160-166
    syn = cfg.data['synthetic']
    loaders = [SyntheticLoader(s['length'], cfg.get('videos_per_gpu', 1), s.get('frames', 8), s.get('tokens', 32),
                               1000 * (i + 1) + rank, dev) for i, s in enumerate(syn)]
    opt, lrc = cfg.optimizer, cfg.lr_config
    engine = CloverEngine(model, next(iter(loaders[0])), lr=lr, betas=tuple(opt.get('betas', (0.9, 0.999))),
                          eps=opt.get('eps', 1e-8), weight_decay=opt.get('weight_decay', 0.0),
                          paramwise_cfg=opt.get('paramwise_cfg'),
                          grad_clip=(cfg.get('optimizer_config', {}).get('grad_clip') or {}).get('max_norm', 0.0),
                          # `fp16 = dict(loss_scale=...)` of a reference config (pretrain_webvid_cc3m.py:21) -> the engine's
                          # device-resident loss scaler (fp16 build; the bf16 build keeps its default: none)
                          loss_scale=(cfg.get('fp16') or {}).get('loss_scale') if clover_amd._lib.HALF_F16 else None)
    if lrc.get('policy', 'CosineAnnealing') != 'CosineAnnealing' or lrc.get('by_epoch', True):
        raise NotImplementedError('lr_config: only CosineAnnealing with by_epoch=False (the reference recipe)')
    if cfg.get('hip_graph', True):                       # static shapes: replay the step as hipGraphs
        first = next(iter(loaders[0]))
        engine.dry_step(first)                           # no optimizer step: training starts from the initial weights
        engine.capture(first)
    runner = CloverRunner(engine, model=model, work_dir=cfg.work_dir, max_epochs=cfg.total_epochs,
                          meta=dict(config_name=os.path.basename(args.config), seed=args.seed))
    # the LR follows runner.iter (batch indices), not the count of optimizer steps (two per index with two loaders)
    runner.register_hook(LrUpdaterHook(lr, **{k: v for k, v in dict(lrc).items() if k not in ('policy', 'by_epoch')}))
    if rank == 0:
        runner.register_hook(LogHook(cfg.get('log_config', {}).get('interval', 10), printer=print))
        if cfg.get('checkpoint_config'):
            runner.register_hook(CheckpointHook(cfg.work_dir, cfg.checkpoint_config.get('interval', 1)))
    if args.resume_from:
        runner.resume(args.resume_from)
    elif args.load_from:
        runner.load_checkpoint(args.load_from)
    runner.run(loaders, [tuple(w) for w in cfg.get('workflow', [('train', 1)])], cfg.total_epochs)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
