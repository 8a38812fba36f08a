"""Wall time of the step's phases in hipGraph mode: forward graph, eager cross-rank section, backward graph(s),
optimizer."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench, clover_amd
from clover_amd.engine import CloverEngine
dev = torch.device('cuda', 0)
import torch.distributed as dist
if os.environ.get('CLOVER_FORCE_COLLECTIVES') == '1':
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=dev, rank=0, world_size=1)
torch.manual_seed(1234)
cfg = bench.model_cfg('T', 8)
if os.environ.get('TEXT_LAYERS'):          # diagnostic only: is the text tower (side stream) on the critical path of a phase?
    cfg['text_backbone']['num_hidden_layers'] = int(os.environ['TEXT_LAYERS'])
if os.environ.get('SWIN_DEPTH2'):          # ... or the video tower: depth of Swin stage 2
    cfg['backbone']['depths'][2] = int(os.environ['SWIN_DEPTH2'])
model = clover_amd.build_model(cfg).to(dev); model.train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(8, 8, 32, 1000).items()}
eng = CloverEngine(model, batch, lr=1e-5, weight_decay=0.005, grad_clip=15.0, max_iters=100000)
eng.step(batch); eng.capture(batch)
for _ in range(5): eng.step(batch)
torch.cuda.synchronize()
ev = lambda: torch.cuda.Event(enable_timing=True)
acc = [0.0] * 4
N = 20
for _ in range(N):
    e = [ev() for _ in range(5)]
    e[0].record()
    eng.graph.replay()
    e[1].record()
    emb = eng._static_emb.detach().requires_grad_(); mlm = eng._static_mlm.detach().requires_grad_()
    losses = model.contrastive_losses(emb, mlm)
    loss, lv = model._parse_losses(losses)
    loss.backward()
    eng._static_demb.copy_(emb.grad); eng._static_dmlm.copy_(mlm.grad)
    e[2].record()
    eng._replay_backward()
    e[3].record()
    eng.reducer.finish()
    e5 = ev(); e5.record()
    eng.optimizer_step()
    e[4].record()
    torch.cuda.synchronize()
    for i in range(4): acc[i] += e[i].elapsed_time(e[i + 1])
    fin = globals().get('fin', 0.0) + e[3].elapsed_time(e5); globals()['fin'] = fin
print('forward graph %.2f ms | eager losses %.2f ms | backward graph %.2f ms | optimizer %.2f ms | total %.2f' % (
    acc[0] / N, acc[1] / N, acc[2] / N, acc[3] / N, sum(acc) / N))
print('  of the optimizer phase, reducer.finish(): %.2f ms' % (fin / N))
