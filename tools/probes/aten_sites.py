"""Which Python lines of clover_amd issue the ATen ops of one eager step (TorchDispatchMode + traceback; backward kept on the
calling thread): (op, innermost clover_amd frame) -> calls.  Only ops that write device memory matter: views are skipped."""
import sys, os, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench, clover_amd
from clover_amd.engine import CloverEngine
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
model = clover_amd.build_model(bench.model_cfg('T', 8)).to(dev); model.train()
batch = {k: v.to(dev) for k, v in bench.synthetic_batch(8, 8, 32, 1000).items()}
eng = CloverEngine(model, batch, lr=1e-5, weight_decay=0.005, grad_clip=15.0, max_iters=100000)
for _ in range(2): eng.step(batch)
torch.cuda.synchronize()
SKIP = {'view', 'reshape', 'transpose', 'permute', 'slice', 'select', 'expand', 'unsqueeze', 'squeeze', 'as_strided', 't',
        'detach', 'alias', 'unbind', 'split', 'narrow', '_unsafe_view', 'empty', 'empty_like', 'empty_strided', 'sym_size',
        'sym_stride', 'sym_numel', 'sym_storage_offset', 'stride', 'size', 'is_contiguous', 'is_pinned', 'numel', 'dim',
        'unflatten', 'flatten', 'chunk', 'lift_fresh', 'set_', 'resize_', '_local_scalar_dense', 'item', 'new_empty',
        'record_stream', 'view_as', 'split_with_sizes', 'unsafe_split', 'unsafe_chunk', 'prim', 'is_same_size', 'result_type'}
agg = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if name.replace('aten.', '').split('.')[0] not in SKIP:
            site = '?'
            for fr in reversed(traceback.extract_stack(limit=40)):
                if '/clover_amd/' in fr.filename and 'probes' not in fr.filename:
                    site = f"{fr.filename.split('/clover_amd/')[-1]}:{fr.lineno} {fr.name}"
                    break
            if site.startswith('engine.py'):        # issued by autograd's own backward nodes: the operand shapes name the forward op
                site += '  ' + ' '.join(str(tuple(a.shape)) for a in args if torch.is_tensor(a))[:60]
            agg[(name.replace('aten.', ''), site)] += 1
        return func(*args, **(kwargs or {}))


torch.autograd.set_multithreading_enabled(False)
with Rec():
    eng.step(batch)
torch.cuda.synchronize()
print('ops', sum(agg.values()))
for (name, site), n in sorted(agg.items(), key=lambda kv: (-kv[1], kv[0]))[:140]:
    print(f'{n:4d}  {name:34s} {site}')
