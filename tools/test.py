#!/usr/bin/env python
"""tools/test.py-style evaluation driver for the retrieval task (reference: tools/test.py:27-100 arguments,
:130-260 main; collection mmaction/core/hooks/my_eval_hook.py:20-100; metrics video_dataset.py:189-195).

    python tools/test.py configs/finetune_retrieval_synthetic.py work_dirs/.../epoch_2.pth --eval recall_for_video_text_retrieval
    python -m torch.distributed.run --nproc-per-node 8 tools/test.py <config> <checkpoint> --launcher pytorch --eval ...

The model runs ``forward_test(separate_test=True)`` (the HIP Swin + BERT paths, forward only) over a test set sharded
rank-major; embeddings are collected over RCCL and rank 0 computes R@1/5/10, median rank.  The dataset side is out of
scope: ``data.synthetic_test`` describes a synthetic test set (``pairs`` random (clip, caption) pairs, so the metrics
of an untrained model sit at chance: R@K ~ 100 K / pairs)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch                                            # noqa: E402
import torch.distributed as dist                        # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description='test (and eval) a model')
    p.add_argument('config', help='test config file path')
    p.add_argument('checkpoint', help="checkpoint file ('none' = the seeded random init)")
    p.add_argument('--out', default=None, help='output result file (json)')
    p.add_argument('--eval', type=str, nargs='+', default=['recall_for_video_text_retrieval'], help='evaluation metrics')
    p.add_argument('--gpu-collect', action='store_true', help='accepted for CLI compatibility (collection is always RCCL)')
    p.add_argument('--cfg-options', nargs='+', default=[], help='a.b=c overrides merged into the config')
    p.add_argument('--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    return p.parse_args()


class SyntheticTestLoader:
    """This rank's shard of a synthetic test set: batches with ``index`` as the reference's test pipeline emits."""

    def __init__(self, pairs, batch, frames, tokens, rank, world, device, seed=4242):
        import bench
        self.batches = []
        mine = list(range(rank, pairs, world))
        for s in range(0, len(mine), batch):
            idx = mine[s:s + batch]
            b = bench.synthetic_batch(len(idx), frames, tokens, seed + idx[0])
            b = {k: b[k].to(device) for k in ('imgs', 'token_ids', 'segment_ids', 'input_mask')}
            b['index'] = torch.tensor(idx, device=device)
            self.batches.append(b)

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def main():
    args = parse_args()
    from clover_amd.runner import Config, parse_cfg_options
    from clover_amd.evaluation import evaluate_retrieval, multi_gpu_test_retrieval
    import clover_amd
    cfg = Config.fromfile(args.config)
    cfg.merge_from_dict(parse_cfg_options(args.cfg_options))
    if not torch.cuda.is_available():
        raise SystemExit('tools/test.py needs an MI355X (no CPU fallback)')
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
    torch.manual_seed(0)
    model = clover_amd.build_model(cfg.model.copy() if hasattr(cfg.model, 'copy') else dict(cfg.model)).to(dev)
    if args.checkpoint != 'none':
        ckpt = torch.load(args.checkpoint, map_location='cpu')
        sd = ckpt.get('state_dict', ckpt)
        sd = {(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()}
        res = model.load_state_dict(sd, strict=False)
        if rank == 0:
            print(f'loaded {args.checkpoint}: {len(res.missing_keys)} missing, {len(res.unexpected_keys)} unexpected keys')
    model.eval()
    st = cfg.data.get('synthetic_test', dict(pairs=64, frames=8, tokens=32))
    loader = SyntheticTestLoader(st.get('pairs', 64), cfg.get('videos_per_gpu', 8), st.get('frames', 8),
                                 st.get('tokens', 32), rank, world, dev)
    results = multi_gpu_test_retrieval(model, loader)
    if rank == 0:
        metrics = evaluate_retrieval(results, args.eval)
        for k, v in metrics.items():
            print(f'{k}: {v:.04f}')                                                   # tools/test.py:255-256
        if args.out:
            with open(args.out, 'w') as f:
                json.dump(dict(metrics={k: float(v) for k, v in metrics.items()}, pairs=int(len(results['index']))), f)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
