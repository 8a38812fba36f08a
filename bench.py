"""Headline benchmark: video-text pairs/sec of the full Clover pre-training step
(forward + backward + gradient all-reduce + clip + AdamW) on N MI355X of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 5

Workload = BASELINE.json configs[1] (N=1) / configs[2] (N>1): VideoSwin-T + BERT-base + 3-layer
fusion, per-GPU batch 8 clips x 8 frames x 224^2 + 32-token captions, all five losses on, bf16
MFMA compute with fp32 master weights.  Synthetic data (SURVEY §8d generator, seed 1000+rank),
random-init weights, inputs resident in HBM before the timed region.  Weak scaling: per-GPU
batch fixed; `value` = global pairs / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

BERT_BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, layer_norm_eps=1e-12)
AUX = ['token_ids', 'segment_ids', 'input_mask', 'mlm_label', 'v_token_mask']
GF_PER_PAIR = {('T', 8): 334.4, ('B', 8): 885.2, ('B', 16): 1846.0, ('B', 32): 3663.2}   # BASELINE.md §2, fwd+bwd


def model_cfg(variant='T', frames=8):
    """configs/exp_local/pretrain_webvid_cc3m.py:22-112 with the Swin-T backbone of
    configs/_base_/models/swin3d/swin3d_tiny.py (BASELINE configs[1])."""
    swin = dict(T=dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], drop_path_rate=0.1),
                B=dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], drop_path_rate=0.3))[variant]
    cf = swin['embed_dim'] * 8
    return dict(
        type='CloverPretrain', freeze_stage=None, separate_test=True, use_Cmask=True,
        backbone=dict(type='SwinTransformer3D', patch_size=(2, 4, 4), stride=(2, 4, 4), window_size=(8, 7, 7),
                      mask_token=True, pretrained2d=False, pretrained=None, **swin),
        freeze_text_backbone=None, text_vocab_size=30522,
        mm_backbone=dict(type='CrossModalTransformerFromPretrained', use_text_cls=True, use_prompt=False,
                         pretrained_model='bert-base-uncased', num_hidden_layers=3, img_in_size=cf, hidden_size=768,
                         num_frames=max(1, (frames + 1) // 2), spacial_tokens=49, token_types=2, layer_norm_eps=1e-12,
                         word_pos_start=False, bert_config=dict(BERT_BASE)),
        text_backbone=dict(type='BertFromPretrained', num_hidden_layers=12, bert_config=dict(BERT_BASE)),
        cls_head=None,
        ssl_head=dict(type='NCEHeadForMM', visual_in_channels=cf, text_in_channels=768, img_hidden_dim=1536,
                      vts_embed_dim=768, ln=True, spatial_type='avg', text_agg_type='cls', dropout_ratio=0),
        mlm_head=dict(type='MLMHead', hidden_size=768, vocab_size=30522),
        mlm_ssl_head=dict(
            V=dict(type='NCEHeadForVision', visual_in_channels=768, cross_in_channels=768, hidden_dim=768, ln=True,
                   vts_embed_dim=768, dropout_ratio=0),
            T=dict(type='NCEHeadForText', cross_in_channels=768, vts_embed_dim=768, text_bn=False, dropout_ratio=0.1)),
        mlm_loss=dict(type='SoftmaxFocalLossMultiClass', gamma=2.0),
        loss_type=dict(type='CrossEntropyLoss'),
        ssl_loss=dict(type='ExclusiveNCEwithRankingLoss', temperature=0.05, use_rank=True, use_rank_ttm=True,
                      use_rank_trtm=False, margin_ttm=5., margin_trtm=10.),
        symmetry_rank=True, train_cfg=dict(aux_info=list(AUX)))


def synthetic_batch(B, frames, L, seed, size=224):
    """SURVEY §8d generator: N(0,1) clips; [CLS] ids [SEP] pad; 30 % MLM positions (80 % -> [MASK]);
    block-wise ~10-cell 7x7 video mask."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(B, 1, 3, frames, size, size, generator=g)
    ids = torch.zeros(B, 1, L, dtype=torch.long)
    mlm = torch.full((B, 1, L), -100, dtype=torch.long)
    vm = torch.zeros(B, 1, 7, 7, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(6, L - 1, (1,), generator=g))
        ids[b, 0, 0] = 101
        ids[b, 0, 1:1 + n] = torch.randint(1000, 30000, (n,), generator=g)
        ids[b, 0, 1 + n] = 102
        sel = torch.rand(n, generator=g) < 0.3
        sel[0] = True
        pos = torch.nonzero(sel).flatten() + 1
        mlm[b, 0, pos] = ids[b, 0, pos]
        to_mask = pos[torch.rand(len(pos), generator=g) < 0.8]
        ids[b, 0, to_mask] = 103
        r0, c0 = int(torch.randint(0, 5, (1,), generator=g)), int(torch.randint(0, 4, (1,), generator=g))
        vm[b, 0, r0:r0 + 3, c0:c0 + 4] = 1
        vm[b, 0, r0, c0] = 0
        vm[b, 0, r0 + 2, c0 + 3] = 0
    return dict(imgs=imgs, label=torch.zeros(B, dtype=torch.long), token_ids=ids,
                segment_ids=torch.zeros_like(ids), input_mask=(ids != 0).long(), mlm_label=mlm, v_token_mask=vm)


def oracle_cfg(cfg):
    bb = {k: cfg['backbone'][k] for k in ('patch_size', 'embed_dim', 'depths', 'num_heads', 'window_size')}
    return dict(backbone=bb,
                bert=dict(num_hidden_layers=cfg['text_backbone']['num_hidden_layers'], num_attention_heads=12,
                          layer_norm_eps=1e-12),
                fusion=dict(num_hidden_layers=cfg['mm_backbone']['num_hidden_layers'], num_attention_heads=12,
                            layer_norm_eps=1e-12),
                temperature=0.05, margin=5.0, gamma=2.0, vocab=30522)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(cfg, state, frames, L, budget_s=25.0):
    """The oracle (fp32 CPU restatement of the reference step, parity-pinned to the reference by
    tests/golden) timed on this box's host cores: forward + losses + backward, B = 2."""
    from oracle import model as om
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    P = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())
         for k, v in state.items() if 'relative_position_index' not in k}
    B = 2
    batch = synthetic_batch(B, frames, L, seed=999)
    ocfg = oracle_cfg(cfg)
    times = []
    t_start = time.time()
    warm = None
    ref_losses = None
    for it in range(4):
        t0 = time.time()
        losses = om.forward_train(P, batch, ocfg, gather=False)
        loss, lv = om.parse_losses(losses)
        if ref_losses is None:
            ref_losses = {k: float(v) for k, v in lv.items()}
        loss.backward()
        dt = time.time() - t0
        if it > 0:
            times.append(dt)
        else:
            warm = dt
        for p in P.values():
            p.grad = None
        if time.time() - t_start > budget_s:
            break
    note = 'timed iters after 1 warm-up'
    if not times:                       # slow host: the warm-up iteration is all the budget allows
        times, note = [warm], 'iter (the warm-up itself; budget exhausted)'
    per = sum(times) / len(times)
    # kind "port": the oracle restatement, not the reference itself (which cannot travel to the GPU box).  Same cores, same
    # weights / batch, the two differ by -20 % .. +30 % (oracle / reference speed 0.81-1.15 at these shapes,
    # 1.28-1.32 at Swin-B / 32 f)
    return dict(value=round(B / per, 4), unit='pairs/s', cores=ncores, kind='port', losses=ref_losses,
                port_speed_over_reference='0.81-1.15 at config-2 shapes, 1.28-1.32 at Swin-B 32 f (BASELINE.md section 5)',
                sample=f'oracle forward_train+backward, fp32, B={B}, {frames}f x 224^2, L={L}, '
                       f'{len(times)} {note} ({per:.2f} s/iter)')


def cpu_baseline_subprocess(variant, frames, L, timeout_s=240):
    """Run the CPU leg in a child process that never touches the GPU (own thread pool, hard
    timeout) so it cannot stall the GPU process; same seed -> same random-init weights."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--variant', variant,
           '--frames', str(frames), '--tokens', str(L)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return dict(value=None, unit='pairs/s', cores=usable_cores(), kind='port',
                    sample='cpu leg failed: ' + (out.stderr.strip().splitlines() or ['?'])[-1][:200])
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='pairs/s', cores=usable_cores(), kind='port',
                    sample=f'cpu leg exceeded {timeout_s} s and was stopped')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch', type=int, default=8, help='clips per GPU')
    ap.add_argument('--frames', type=int, default=8)
    ap.add_argument('--tokens', type=int, default=32)
    ap.add_argument('--variant', default='T', choices=['T', 'B'])
    ap.add_argument('--loss-scale', default=None, help="probe: 'dynamic', a number (static) or 'none'; default: dynamic from 1024 for f16, none for bf16")
    ap.add_argument('--dtype', default=os.environ.get('CLOVER_BENCH_DTYPE', 'f16'), choices=['f16', 'bf16', 'fp8'],
                    help='f16 (default): activations, weight shadows and gradients in IEEE fp16 with a static loss scale — the '
                         'reference trains in fp16 (pretrain_webvid_cc3m.py:21); bf16: the same kernels compiled for bf16 '
                         '(libclover_hip.so: same step time, losses 1.8e-2 instead of 2e-3 from the fp32 reference); '
                         'fp8: forward GEMMs on e4m3 operands (BASELINE config 5) over bf16, gradients stay bf16')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--phases', action='store_true', help='after the timed region: GPU time per phase of the graphed step (events at the phase boundaries), as a "phases_ms" field')
    ap.add_argument('--no-graph', action='store_true', help='run the step eagerly instead of as one hipGraph')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_only:                           # child of cpu_baseline_subprocess(): CPU only
        import clover_amd
        torch.manual_seed(1234)
        cfg = model_cfg(args.variant, args.frames)
        model = clover_amd.build_model(cfg)
        state = {k: v.detach() for k, v in model.state_dict().items()}
        print(json.dumps(cpu_baseline(cfg, state, args.frames, args.tokens)))
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    if rank != 0:                                        # only rank 0 reports: keep the other ranks' library banners off stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1 or os.environ.get('CLOVER_FORCE_COLLECTIVES') == '1':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        dist.init_process_group('nccl', device_id=dev, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    # read when clover_amd is first imported: selects the build of the kernels (fp8 mode quantises bf16 tensors)
    os.environ['CLOVER_HALF'] = 'f16' if args.dtype == 'f16' else 'bf16'
    import clover_amd
    from clover_amd import ops
    from clover_amd.engine import CloverEngine
    if args.dtype == 'fp8':
        ops.FP8 = True

    torch.manual_seed(1234)                              # identical init on every rank (== DDP broadcast)
    cfg = model_cfg(args.variant, args.frames)
    model = clover_amd.build_model(cfg).to(dev)
    model.train()                                        # dropout / DropPath active, as in training

    # Loss parity of THIS model (the weights the CPU leg rebuilds from the same seed) before any training step: the
    # B = 2 batch the oracle is timed on, eval mode (no dropout / DropPath, as the oracle), on the bf16 training path and
    # in parity mode (fp32 storage + arithmetic on the HIP kernels); compared with the oracle's losses below.
    own_losses = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from clover_amd import parity
        cb = {k: v.to(dev) for k, v in synthetic_batch(2, args.frames, args.tokens, seed=999).items()}
        model.eval()
        with torch.no_grad():
            fast = {k: float(v) for k, v in model.train_step(cb, None)['log_vars'].items()}
            with parity.mode():
                par = {k: float(v) for k, v in model.train_step(cb, None)['log_vars'].items()}
        own_losses = {args.dtype: fast, 'parity': par}      # the timed build's kernels | the fp32 instantiation of the path
        model.train()
        del cb

    batch = {k: v.to(dev) for k, v in synthetic_batch(args.batch, args.frames, args.tokens, 1000 + rank).items()}
    # loss scaling as the headline config has it (pretrain_webvid_cc3m.py:21 fp16 = dict(loss_scale='dynamic')): the dynamic
    # scaler on the device, started at the scale the reference's settles near instead of 2**32 — the first ~20 steps of a
    # 2**32 start overflow and are skipped, and a skipped step is not a measured step ("steps_skipped" below must be 0)
    scaler = dict(init_scale=1024.0, mode='dynamic') if args.dtype == 'f16' else None
    if args.loss_scale is not None:            # probe: 'dynamic' (as above), a number (static), 'none' (1.0: no scaler at all)
        scaler = (dict(init_scale=1024.0, mode='dynamic') if args.loss_scale == 'dynamic'
                  else 1.0 if args.loss_scale == 'none' else float(args.loss_scale))
    engine = CloverEngine(model, batch, lr=5e-5 / 1024 * args.batch * world, weight_decay=0.005, grad_clip=15.0,
                          max_iters=100000, loss_scale=scaler)
    # GEMMs that left the own kernels BEFORE the measured step exists (the eval-mode loss check above on the un-managed
    # model, the engine's parameter census: both run the MLM decoder before its vocabulary rows are padded in the slab)
    # are reported apart from those of the step itself (eager step, capture, warm-up, timed region)
    lib_setup = dict(ops.LIBRARY_GEMM_CALLS)
    ops.LIBRARY_GEMM_CALLS.clear()

    def sync():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    graphed = False
    if not args.no_graph:
        engine.step(batch)                               # one eager step: lazy inits, kernel attributes
        graphed = engine.capture(batch)
        if graphed:
            # the synthetic batch lives in the graphs' static input buffers (where a prefetching loader would put it):
            # no device-to-device staging copy per step
            batch = engine.input_buffers()
    for _ in range(args.warmup):
        out = engine.step(batch)
    sync()
    if not args.no_kernel_timing and not graphed:
        ops.PROF = {}
    if engine.reducer.active:
        engine.reducer.start_timing()                    # stall of the compute stream on the gradient all-reduces
    skipped0 = ops.optim_state_read(engine.optim_state)['skipped']
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = engine.step(batch)
    sync()
    dt = time.perf_counter() - t0
    steps_skipped = ops.optim_state_read(engine.optim_state)['skipped'] - skipped0      # overflow-skipped updates in the timed region
    exposed_comm_ms = engine.reducer.exposed_ms() if engine.reducer.active else None
    prof, ops.PROF = ops.PROF, None
    prof_steps = args.steps
    with_copy_ms = None
    if graphed:
        # the same steps fed from tensors OUTSIDE the graphs' static input buffers (what a loader that does not write
        # into engine.input_buffers() pays: one device-to-device staging copy of the batch per step) — ADVICE r2
        fresh = {k: v.clone() for k, v in batch.items()}
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out = engine.step(fresh)
        sync()
        with_copy_ms = (time.perf_counter() - t1) / args.steps * 1e3
        del fresh
    phases_ms = None
    if graphed and args.phases:
        engine.start_phase_timing()
        for _ in range(args.steps):
            engine.step(batch)
        sync()
        phases_ms = {k: round(v, 3) for k, v in engine.phase_ms().items()}
    if graphed and not args.no_kernel_timing:
        # per-kernel durations: HIP events cannot bracket launches inside a replayed graph, so the same
        # kernels (same shapes, same data) are timed on 3 eager steps right after the timed region.  Each
        # eager step is queued behind ~40 ms of filler GEMMs so that the host runs ahead and the step's
        # kernels execute back-to-back at full clocks, as they do inside the graph.
        engine.graph, g = None, engine.graph
        model.overlap_text = False        # serial streams while single kernels are bracketed by events (as rocprof sees them)
        filler = torch.randn(8192, 8192, device=dev, dtype=ops.BF16)
        ops.PROF = {}
        prof_steps = 3
        for _ in range(prof_steps):
            for _ in range(40):
                torch.mm(filler, filler)
            engine.step(batch)
        sync()
        del filler
        prof, ops.PROF = ops.PROF, None
        engine.graph = g
        model.overlap_text = True
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    log_vars = {k: float(v) for k, v in out['log_vars'].items()}
    gnorm = engine.grad_norm()

    if rank == 0:
        gb = args.batch * world
        pairs_s = gb * args.steps / dt
        gf = GF_PER_PAIR.get((args.variant, args.frames))
        res = {
            'metric': f'video-text pairs/sec ({args.frames}f x 224^2, {args.tokens}-tok), full pre-training step',
            'value': round(pairs_s, 3), 'unit': 'pairs/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic', 'hip_graph': graphed,
            'config': {'workload': f'VideoSwin-{args.variant} + BERT-base + 3-layer fusion, MLM + tri-modal '
                                   f'exclusive InfoNCE + rank losses, {args.frames}f x 224^2, {args.tokens}-tok',
                       'per_gpu_batch': args.batch, 'global_batch': gb, 'parallelism': f'dp{world}',
                       'params_M': round(engine.num_params / 1e6, 1),
                       'first_touch_params_M': round(getattr(engine, 'first_touch_params', 0) / 1e6, 1)},
            'step_tflops': round(pairs_s * gf / 1e3, 2) if gf else None,
            'frac_bf16_mfma_peak': round(pairs_s * gf / 1e3 / (2500.0 * world), 4) if gf else None,      # (fp16 MFMA: the same dense peak)
            'losses': {k: round(v, 4) for k, v in log_vars.items()}, 'grad_norm': round(gnorm, 4),
            'loss_scale': engine.loss_scaler_state(), 'steps_skipped': steps_skipped,
            'peak_mem_GB': round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
            # data-parallel runs: mean per-step stall of the compute stream on the gradient all-reduces (rank 0), and what
            # travels: bf16 gradients in per-class buckets (null at N = 1: no collective is issued)
            'exposed_comm_ms': round(exposed_comm_ms, 3) if exposed_comm_ms is not None else None,
            'phases_ms': phases_ms,
            'grad_allreduce_dtype': ('bf16' if engine.wire is not None else 'fp32') if engine.reducer.active else None,
        }
        if prof:
            res['roofline'], res['kernels'], roof2 = ops.roofline_from_prof(prof, prof_steps)
            if roof2 is not None:
                res['roofline_2'] = roof2
            # Counter-derived fields cannot be read from inside the process: they come from the committed rocprofv3 passes of
            # THIS command (tools/pmc_traffic.sh, tools/pmc_mfma.sh, tools/gpu_round.sh), each file carrying the commit it
            # was taken at; the *_source fields say which file and commit, so a reader can tell a fresh counter from a stale one.
            prof_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from csrc_hash import csrc_sha16
            cur_sha = csrc_sha16(ROOT)

            def newest(suffix):
                """profiles/rNN_<suffix> of the highest round NN present"""
                import re
                best = None
                for f in os.listdir(prof_dir) if os.path.isdir(prof_dir) else []:
                    m = re.match(r'r(\d\d)_' + re.escape(suffix) + '$', f)
                    if m and (best is None or int(m.group(1)) > best[0]):
                        best = (int(m.group(1)), f)
                return best[1] if best else None

            def clean(nm):
                return nm.replace('(anonymous namespace)::', '').replace('void ', '')

            def counter_fields(roof):
                """traffic / mfma_util / avg_us_rocprof of one roofline block from the committed rocprofv3 files; kernels are
                matched by PREFIX of the cleaned name (the PMC json keys carry template arguments the event keys do not:
                `adamw_dev_kernel<float>` — VERDICT r5 Weak 12)."""
                kname = roof['kernel']
                stale = []

                def committed(suffix):
                    fname = newest(suffix)
                    if fname is None:
                        return None, None
                    d = json.load(open(os.path.join(prof_dir, fname)))
                    meta = d.get('_meta') or {}
                    if meta.get('csrc_sha16') != cur_sha:
                        stale.append(fname)                      # taken from other kernel sources than this tree's
                    rec = d.get(kname)
                    if rec is None:
                        hits = [v for k, v in d.items() if k != '_meta' and clean(k).startswith(kname)]
                        rec = hits[0] if len(hits) == 1 else None
                    return rec, f"profiles/{fname}@{meta.get('commit', 'unknown')}"
                rec, src = committed('pmc_traffic.json')
                roof['traffic'] = rec['hbm_bytes_per_launch'] if rec else None
                roof['traffic_source'] = src if rec else None
                rec, src = committed('pmc_mfma.json')
                roof['mfma_util'] = rec['mfma_util'] if rec else None
                roof['mfma_util_source'] = src if rec else None
                # the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary of this command
                # (device-side, no event / dispatch overhead) and the roofline fraction it gives
                sname = newest('kernel_stats_bench_default_final.csv')
                if sname:
                    import csv
                    meta = os.path.join(prof_dir, sname.replace('.csv', '.meta.json'))
                    md = json.load(open(meta)) if os.path.exists(meta) else {}
                    if md.get('csrc_sha16') != cur_sha:
                        stale.append(sname)
                    for row in csv.DictReader(open(os.path.join(prof_dir, sname))):
                        if clean(row['Name']).startswith(kname):
                            us = float(row['AverageNs']) / 1e3
                            roof['avg_us_rocprof'] = round(us, 2)
                            per = roof['algorithmic_bytes_per_launch' if roof['bound'] == 'hbm'
                                       else 'algorithmic_flops_per_launch']
                            peak = 8.0e12 if roof['bound'] == 'hbm' else 2.5e15
                            roof['frac_rocprof'] = round(per / (us * 1e-6) / peak, 4)
                            roof['rocprof_source'] = f"profiles/{sname}@{md.get('commit', 'unknown')}"
                            break
                # stale: a committed counter file was taken from kernel sources other than this tree's (csrc hash mismatch): its
                # numbers describe an older build of the kernels — the live HIP-event fields above do not depend on it
                roof['stale'] = bool(stale)
                roof['stale_files'] = sorted(set(stale))
                roof['csrc_sha16'] = cur_sha
            counter_fields(res['roofline'])
            if roof2 is not None:
                counter_fields(res['roofline_2'])
        res['library_gemm_calls'] = {f'{k[0]} {list(k[1])}': v for k, v in ops.LIBRARY_GEMM_CALLS.items()}
        res['library_gemm_calls_setup'] = {f'{k[0]} {list(k[1])}': v for k, v in lib_setup.items()}
        if not args.no_cpu_baseline and world == 1:
            res['cpu_baseline'] = cpu_baseline_subprocess(args.variant, args.frames, args.tokens)
            ref = res['cpu_baseline'].pop('losses', None)
            if ref and own_losses:
                # |loss - oracle| on identical weights / inputs (B = 2, eval): north-star bound 1e-3 for the parity mode
                res['loss_abs_err_vs_oracle'] = {
                    mode: {k: round(abs(v - ref[k]), 7) for k, v in lv.items() if k in ref}
                    for mode, lv in own_losses.items()}
                res['loss_abs_err_vs_oracle']['sample'] = 'B=2 synthetic batch (seed 999), eval mode, seed-1234 init'
                # what each row is: the dtype's name = the TIMED training path (MFMA kernels, 16-bit storage of that build);
                # "parity" = the fp32 instantiation of the path (clover_amd/parity.py: fp32 GEMM / attention kernels through the
                # same index logic) — a check of structure and index logic at the north-star 1e-3, not a bound on the 16-bit kernels
                res['loss_abs_err_vs_oracle']['note'] = (f'{args.dtype} = the timed path; parity = fp32 storage + arithmetic '
                                                         'variant (structure / index-logic check, not the 16-bit kernels)')
        if with_copy_ms is not None:
            res['ms_per_step_with_input_copy'] = round(with_copy_ms, 3)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes a version banner through C stdio: push it out first so that the JSON line is the LAST line of stdout
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
