"""Video-text retrieval metrics of the reference's evaluation stage
(mmaction/core/evaluation/accuracy.py:430-462, ``normalize_fn`` mmaction/utils/numpy_norm.py:5-8).

This is the HOST post-processing the reference's dataset ``evaluate()`` runs in numpy once per epoch on the
embeddings ``forward_test(separate_test=True)`` returned (N x D, N = test-set size, MSRVTT: 1000); it stays host
code here too — the device work is the two encoders that produce the embeddings.
"""
import numpy as np


def normalize_fn(x, axis=-1, order=2):
    """Rows scaled to unit L2 norm; all-zero rows are left untouched (numpy_norm.py:5-8)."""
    x = np.asarray(x)
    l2 = np.atleast_1d(np.linalg.norm(x, ord=order, axis=axis))
    l2[l2 == 0] = 1
    return x / np.expand_dims(l2, axis=axis)


def recall_for_video_text_retrieval(video_embd=None, text_embd=None, input_scores=None, use_sim=False, texts=None):
    """R@1 / R@5 / R@10 (percent), median rank (1-based) and ``Recall@all`` = R1 + R5 + R10 - MR of
    text -> video retrieval; query i's ground truth is video i.  ``use_sim`` is accepted and ignored, as in the
    reference (accuracy.py:438 overwrites it with False); ``texts`` is unused there too."""
    if input_scores is not None:
        scores = np.asarray(input_scores)
    else:
        scores = np.dot(normalize_fn(_host(text_embd)), normalize_fn(_host(video_embd)).T)
    order = np.argsort(-scores, axis=1)
    gt = np.arange(len(scores))
    ind = np.where(order == gt[:, None])[1]
    metrics = {
        'Recall@1': float(np.sum(ind == 0)) / len(ind) * 100,
        'Recall@5': float(np.sum(ind < 5)) / len(ind) * 100,
        'Recall@10': float(np.sum(ind < 10)) / len(ind) * 100,
        'MR': np.median(ind) + 1,
    }
    metrics['Recall@all'] = metrics['Recall@1'] + metrics['Recall@5'] + metrics['Recall@10'] - metrics['MR']
    return metrics


def _host(x):
    if hasattr(x, 'detach'):
        x = x.detach().float().cpu().numpy()
    return np.asarray(x)


def multi_gpu_test_retrieval(model, data_loader, gpu_collect=True):
    """Embed a test set with ``forward_test(separate_test=True)`` on every rank and collect the embeddings on all
    ranks in dataset order (mmaction/core/hooks/my_eval_hook.py:20-100).  Each batch carries ``index`` (the
    samples' positions in the test set); several clips per sample are averaged, several captions per video are
    grouped, as there (:58-63).  Returns ``dict(video_embd=[N x D], text_embd=[N x ...])`` of numpy arrays.

    Collection is one ``all_gather`` of the stacked per-rank embeddings (+ indices) instead of the reference's
    pickle-through-uint8-tensor exchange; a 1-rank run has nothing to collect."""
    import torch
    import torch.distributed as dist
    was_training = model.training
    model.eval()
    vids, txts, idxs = [], [], []
    with torch.no_grad():
        for data in data_loader:
            data = dict(data)
            idxs.append(data.pop('index').reshape(-1).to(torch.int64))
            data.pop('img_metas', None)
            data.pop('label', None)
            v, t = model(return_loss=False, **data)
            if v.shape[0] > t.shape[0]:                                               # :58-60
                v = v.view(t.shape[0], -1, t.shape[1]).mean(dim=1)
            elif v.shape[0] < t.shape[0]:                                             # :61-63 (multiple choice)
                t = t.view(v.shape[0], -1, t.shape[1])
            vids.append(v.float())
            txts.append(t.float())
    model.train(was_training)
    v, t, ix = torch.cat(vids), torch.cat(txts), torch.cat(idxs).to(vids[0].device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        W = dist.get_world_size()
        n = torch.tensor([v.shape[0]], device=v.device)
        ns = [torch.zeros_like(n) for _ in range(W)]
        dist.all_gather(ns, n)
        mx = int(max(x.item() for x in ns))

        def gather(x):
            pad = x.new_zeros((mx,) + tuple(x.shape[1:]))
            pad[:x.shape[0]] = x
            out = [torch.empty_like(pad) for _ in range(W)]
            dist.all_gather(out, pad)
            return torch.cat([o[:int(k.item())] for o, k in zip(out, ns)])
        v, t, ix = gather(v), gather(t), gather(ix)
    # dataset order; a DistributedSampler pads the last ranks with repeated samples: keep the first of each index
    order = torch.argsort(ix, stable=True)
    ix, v, t = ix[order], v[order], t[order]
    keep = torch.ones_like(ix, dtype=torch.bool)
    keep[1:] = ix[1:] != ix[:-1]
    return dict(video_embd=v[keep].cpu().numpy(), text_embd=t[keep].cpu().numpy(), index=ix[keep].cpu().numpy())


def evaluate_retrieval(results, metrics=('recall_for_video_text_retrieval',)):
    """The retrieval branch of ``VideoDataset.evaluate`` (mmaction/datasets/video_dataset.py:189-195)."""
    out = {}
    for metric in ([metrics] if isinstance(metrics, str) else metrics):
        if metric != 'recall_for_video_text_retrieval':
            raise KeyError(f'metric {metric} is not supported')                      # video_dataset.py:163-165
        out.update(recall_for_video_text_retrieval(np.stack(list(results['video_embd'])),
                                                   np.stack(list(results['text_embd']))))
    return out
