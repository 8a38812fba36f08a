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
