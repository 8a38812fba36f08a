"""world_size-2 (gloo, CPU) coverage of the N>1 path: the feature all-gather with its local-slice
backward (reference R6), the packed single-collective variant, unequal batches, the bucketed
gradient all-reduce, the logged-loss averaging — and the oracle's distributed path against the
reference's own 2-rank goldens (g_dist.npz)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _run(fn, world, port, *args):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=fn, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda r: r[0])


# ----------------------------------------------------------------------------- gather + R6
def _w_gather(rank, world, port, q):
    _init(rank, world, port)
    from clover_amd.utils.gather_loss import GatherLoss, VariedShapeGatherLoss, packed_all_gather
    from oracle import model as om
    import closed_form as cf
    G, D = 4, 32
    full = [cf.cf_float(f'dg.e{k}', (G, D), 1.0) for k in range(4)]
    per = G // world
    loc = [f[rank * per:(rank + 1) * per].clone().requires_grad_() for f in full]
    g = packed_all_gather(loc)
    ok_order = all(torch.equal(a.detach(), b) for a, b in zip(g, full))            # rank-major concat
    l = om.exclusive_nce_rank_loss(*g, gather=False)
    (l['nce_loss'] + l['rank_t_tm_loss']).backward()
    # single-process reference on the concatenated batch
    ref = [f.clone().requires_grad_() for f in full]
    lr = om.exclusive_nce_rank_loss(*ref, gather=False)
    (lr['nce_loss'] + lr['rank_t_tm_loss']).backward()
    ok_loss = abs(l['nce_loss'].item() - lr['nce_loss'].item()) < 1e-6
    # backward = LOCAL SLICE of the full gradient, no reduction over ranks (R6)
    ok_grad = all(torch.allclose(a.grad, b.grad[rank * per:(rank + 1) * per], atol=1e-6) for a, b in zip(loc, ref))
    # plain GatherLoss and the varied-shape path with unequal batches
    t = torch.full((2, 3), float(rank)).requires_grad_()
    gl = GatherLoss.apply(t, rank, world)
    ok_gl = gl.shape == (2 * world, 3) and float(gl[2 * rank, 0]) == rank
    n = 1 + rank
    v = torch.full((n, 3), float(rank + 1)).requires_grad_()
    gv = VariedShapeGatherLoss.apply(v, rank, world, False)
    gv.sum().backward()
    ok_var = gv.shape[0] == sum(1 + r for r in range(world)) and torch.equal(v.grad, torch.ones_like(v)) \
        and float(gv[0, 0]) == 1.0 and float(gv[-1, 0]) == float(world)
    q.put((rank, ok_order, ok_loss, ok_grad, ok_gl, ok_var))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_all_variants_world2():
    for r in _run(_w_gather, 2, 29711):
        assert all(r[1:]), r


# ----------------------------------------------------------------------------- bucketed grad all-reduce + log vars
def _w_reducer(rank, world, port, q):
    _init(rank, world, port)
    from clover_amd.utils.grad_reducer import BucketedGradReducer
    from clover_amd.recognizers.base import BaseRecognizer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
    params = list(reversed(list(net.parameters())))
    offs = [0]
    for p in params:
        offs.append(offs[-1] + (p.numel() + 3) // 4 * 4)
    flat = torch.zeros(offs[-1])
    for p, o in zip(params, offs):
        p.grad = flat[o:o + p.numel()].view_as(p)
    red = BucketedGradReducer([(flat, params, offs)], bucket_bytes=256)      # several small buckets
    x = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + rank))

    def loss_of(m, xx):
        # the first layer's weight is used TWICE (tied, like BERT's word embedding / MLM decoder): its gradient
        # arrives in two contributions and the bucket must wait for the second
        return m(xx).pow(2).sum() + (xx @ m[0].weight.t()).tanh().sum()
    for step in range(3):                       # step 0 calibrates the contribution counts, 1.. launch from hooks
        flat.zero_()
        loss_of(net, x).backward()
        red.finish()
    mine = flat.clone()
    # reference: sum over ranks of single-rank grads
    tot = torch.zeros_like(flat)
    for r in range(world):
        net2 = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Tanh(), torch.nn.Linear(16, 4))
        net2.load_state_dict(net.state_dict())
        xr = torch.randn(5, 8, generator=torch.Generator().manual_seed(100 + r))
        loss_of(net2, xr).backward()
        for p, o in zip(reversed(list(net2.parameters())), offs):
            tot[o:o + p.numel()] += p.grad.reshape(-1)
    ok = torch.allclose(mine, tot, atol=1e-5) and len(red.buckets) > 1

    class R(BaseRecognizer):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.lazy_log_vars = True

        def forward_train(self, *a, **k):
            pass

        def forward_test(self, *a, **k):
            pass
    _, lv = R()._parse_losses({'a_loss': torch.tensor(float(rank)), 'b_loss': torch.tensor(2.0)})
    ok_lv = abs(lv['a_loss'] - (world - 1) / 2) < 1e-6 and abs(lv['loss'] - ((world - 1) / 2 + 2.0)) < 1e-6
    q.put((rank, ok, ok_lv))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_and_log_vars_world2():
    for r in _run(_w_reducer, 2, 29712):
        assert all(r[1:]), r


def _w_reducer_wire(rank, world, port, q):
    """bf16 wire slabs + per-class buckets + launch_where: what the engine's graph mode does between backward replays."""
    _init(rank, world, port)
    from clover_amd.utils.grad_reducer import BucketedGradReducer
    torch.manual_seed(0)
    mods = dict(h=torch.nn.Linear(8, 8), t=torch.nn.Linear(8, 8), v1=torch.nn.Linear(8, 8), v0=torch.nn.Linear(8, 4))
    cls_of, params = {}, []
    for c in ('h', 't', 'v1', 'v0'):                       # slab order = the order backward completes the classes
        for p in mods[c].parameters():
            cls_of[id(p)] = c
            params.append(p)
    offs = [0]
    for p in params:
        offs.append(offs[-1] + (p.numel() + 3) // 4 * 4)
    flat = torch.zeros(offs[-1])
    wire = [torch.zeros(offs[-1], dtype=torch.bfloat16)]
    for p, o in zip(params, offs):
        p.grad = flat[o:o + p.numel()].view_as(p)
    red = BucketedGradReducer([(flat, params, offs)], bucket_bytes=1 << 20, split_key=lambda p_: cls_of[id(p_)], wire=wire)
    red.enabled = False                                     # graph mode: the engine launches the buckets itself
    red.reset()
    g = torch.Generator().manual_seed(7 + rank)
    for p in params:
        p.grad.copy_(torch.randn(p.shape, generator=g))
    mine = flat.clone()
    red.begin_step()
    ready = lambda *cs: (lambda p_: cls_of[id(p_)] in cs)
    red.launch_where(ready('h'))
    red.launch_where(ready('h', 't'))
    red.launch_where(ready('h', 'v1'))
    red.finish()
    log = list(red.launch_log)
    ok_cls = [k for _, k, _, _ in log] == ['h', 't', 'v1', 'v0'] and [ph for *_, ph in log] == ['where'] * 3 + ['finish']
    ok_bytes = sum(b for _, _, b, _ in log) == 2 * offs[-1]            # bf16 on the wire: 2 bytes per gradient element
    # reduced values: sum over ranks of the bf16-rounded gradients, in the wire slab; the fp32 slab keeps the local ones
    tot = torch.zeros_like(flat)
    for r in range(world):
        gr = torch.Generator().manual_seed(7 + r)
        for p, o in zip(params, offs):
            tot[o:o + p.numel()] += torch.randn(p.shape, generator=gr).reshape(-1).to(torch.bfloat16).float()
    ok_val = torch.allclose(wire[0].float(), tot, rtol=2 ** -7, atol=1e-6) and torch.equal(flat, mine)
    q.put((rank, ok_cls, ok_bytes, ok_val))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_buckets_leave_per_gradient_class_world2():
    for r in _run(_w_reducer_wire, 2, 29716):
        assert all(r[1:]), r


# ----------------------------------------------------------------------------- oracle vs the reference's 2-rank run
def _w_oracle(rank, world, port, q):
    _init(rank, world, port)
    import closed_form as cf
    import gutil
    from oracle import model as om
    P = {k: v.clone().requires_grad_() for k, v in cf.cf_state(gutil.manifest()).items()}
    batch = cf.cf_batch(4, tag='dist')
    per = 4 // world
    shard = {k: v[rank * per:(rank + 1) * per] for k, v in batch.items()}
    losses = om.forward_train(P, shard, cf.oracle_cfg_from(cf.tiny_model_cfg()), gather=True)
    loss, lv = om.parse_losses(losses)
    loss.backward()
    # DDP semantics: average gradients over ranks
    key = 'backbone.patch_embed.proj.weight'
    g = P[key].grad.clone()
    dist.all_reduce(g)
    g /= world
    q.put((rank, lv, gutil.packed(g)[0] if rank == 0 else None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('W', [2])
def test_oracle_distributed_matches_reference_goldens(W):
    import gutil
    g = gutil.load('g_dist.npz')
    res = _run(_w_oracle, W, 29713 + W)
    lv = res[0][1]
    for k in ['mlm_loss', 'nce_loss', 'rank_t_tm_loss', 'v_nce_loss', 'rank_v_vm_loss', 'loss']:
        assert abs(lv[k] - float(g[f'W{W}.{k}'])) < 2e-4 * max(1, abs(float(g[f'W{W}.{k}']))), (k, lv[k])
    # contrastive losses do not depend on how the global batch is split (W = 1, 2, 4 goldens agree) ...
    for k in ['nce_loss', 'v_nce_loss', 'rank_t_tm_loss', 'rank_v_vm_loss']:
        assert abs(float(g[f'W1.{k}']) - float(g[f'W{W}.{k}'])) < 1e-4 and abs(float(g['W1.' + k]) - float(g['W4.' + k])) < 1e-4
    # ... but the DDP-averaged gradient does: local-slice gather backward (R6)
    sub = res[0][2]
    gsub = g[f'W{W}.grad.backbone.patch_embed.proj.weight.sub'].astype(np.float64)
    assert np.abs(sub - gsub).max() <= 5e-3 * np.abs(gsub).max()
    n1 = g['W1.grad.backbone.patch_embed.proj.weight.stats'][1]
    n2 = g['W2.grad.backbone.patch_embed.proj.weight.stats'][1]
    assert n2 < 0.9 * n1


# ----------------------------------------------------------------------------- retrieval evaluation collection
class _FakeRetrievalModel(torch.nn.Module):
    """Stands in for forward_test(separate_test=True): embeddings are a fixed function of the sample index carried in
    the clip, with `clips` clips per sample (averaged by the collector) — only the collection logic is under test."""

    def forward(self, return_loss=False, imgs=None, token_ids=None, **kw):
        v = imgs.reshape(-1, imgs.shape[-1]).float()                 # [B * clips, D]
        t = token_ids.float()                                        # [B, D]
        return v, t


def _retrieval_loader(rank, world, N, D, clips, batch):
    import closed_form as cf
    V = cf.cf_float('rt.v', (N, D), 1.0)
    T = (0.6 * V + cf.cf_float('rt.t', (N, D), 1.0))
    mine = list(range(rank, N, world))
    if rank == world - 1 and N % world:                                  # sampler-style padding: a repeated sample
        mine.append(0)
    out = []
    for s in range(0, len(mine), batch):
        idx = mine[s:s + batch]
        # clips differ by +-delta so that their mean is the sample's embedding
        d = torch.linspace(-1, 1, clips)[None, :, None] if clips > 1 else torch.zeros(1, 1, 1)
        out.append(dict(imgs=V[idx][:, None, :] + d, token_ids=T[idx], index=torch.tensor(idx)))
    return out, V, T


def _w_retrieval(rank, world, port, q, N, clips):
    _init(rank, world, port)
    from clover_amd.evaluation import evaluate_retrieval, multi_gpu_test_retrieval, recall_for_video_text_retrieval
    loader, V, T = _retrieval_loader(rank, world, N, 16, clips, 3)
    res = multi_gpu_test_retrieval(_FakeRetrievalModel(), loader)
    ok_order = np.array_equal(res['index'], np.arange(N))
    ok_v = np.allclose(res['video_embd'], V.numpy(), atol=1e-6) and np.allclose(res['text_embd'], T.numpy(), atol=1e-6)
    m = evaluate_retrieval(res, ['recall_for_video_text_retrieval'])
    ok_m = m == recall_for_video_text_retrieval(V.numpy(), T.numpy())
    q.put((rank, ok_order, ok_v, ok_m, m['Recall@1']))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('N,clips', [(11, 1), (8, 3)])
def test_retrieval_collection_world2(N, clips):
    """my_eval_hook.py:20-100 — every rank embeds its shard; all ranks end with the whole test set in dataset order,
    sampler padding de-duplicated, multiple clips per sample averaged; metrics equal the single-process ones."""
    res = _run(_w_retrieval, 2, 29571 + N, N, clips)
    for r in res:
        assert all(r[1:4]), r
    assert res[0][4] == res[1][4] and res[0][4] > 30.0               # correlated pairs: far above chance


def test_retrieval_collection_single_process():
    from clover_amd.evaluation import evaluate_retrieval, multi_gpu_test_retrieval
    loader, V, T = _retrieval_loader(0, 1, 9, 16, 2, 4)
    res = multi_gpu_test_retrieval(_FakeRetrievalModel(), loader)
    assert np.array_equal(res['index'], np.arange(9)) and np.allclose(res['video_embd'], V.numpy(), atol=1e-6)
    with pytest.raises(KeyError):
        evaluate_retrieval(res, ['top_k_accuracy'])
