"""ORACLE (test infrastructure only) — integer/index logic of the Clover video encoder.

numpy restatement of the reference's index arithmetic.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product (``clover_amd``) never does.

Pinned against goldens generated from the reference itself
(``tests/golden/make_goldens.py`` -> ``tests/golden/g_idx.npz``), bit-exact.

All citations are relative to /root/reference/.
"""
import numpy as np


def get_window_size(x_size, window_size, shift_size=None):
    """mmaction/models/backbones/swin_transformer_3d.py:302-315.

    Clamp the window to the feature size; zero the shift on clamped axes.
    """
    use_ws = list(window_size)
    use_ss = list(shift_size) if shift_size is not None else None
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            use_ws[i] = x_size[i]
            if use_ss is not None:
                use_ss[i] = 0
    if shift_size is None:
        return tuple(use_ws)
    return tuple(use_ws), tuple(use_ss)


def relative_position_index(window_size):
    """swin_transformer_3d.py:345-359.  [N,N] int64, N = wd*wh*ww, values in
    [0, (2wd-1)(2wh-1)(2ww-1))."""
    wd, wh, ww = window_size
    coords = np.stack(np.meshgrid(np.arange(wd), np.arange(wh), np.arange(ww), indexing='ij'))
    cf = coords.reshape(3, -1)
    rel = cf[:, :, None] - cf[:, None, :]
    rel = rel.transpose(1, 2, 0).copy()
    rel[:, :, 0] += wd - 1
    rel[:, :, 1] += wh - 1
    rel[:, :, 2] += ww - 1
    rel[:, :, 0] *= (2 * wh - 1) * (2 * ww - 1)
    rel[:, :, 1] *= (2 * ww - 1)
    return rel.sum(-1).astype(np.int64)


def window_partition(x, window_size):
    """swin_transformer_3d.py:271-283.  x [B,D,H,W,C] -> [B*nW, N, C]."""
    B, D, H, W, C = x.shape
    wd, wh, ww = window_size
    x = x.reshape(B, D // wd, wd, H // wh, wh, W // ww, ww, C)
    return x.transpose(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, wd * wh * ww, C)


def window_reverse(windows, window_size, B, D, H, W):
    """swin_transformer_3d.py:286-299."""
    wd, wh, ww = window_size
    x = windows.reshape(B, D // wd, H // wh, W // ww, wd, wh, ww, -1)
    return x.transpose(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


def region_ids(D, H, W, window_size, shift_size):
    """Region-id image of compute_mask (swin_transformer_3d.py:551-557), executed
    with the same python slices (slice(-0, None) == whole axis when shift is 0)."""
    img = np.zeros((D, H, W), dtype=np.int32)
    cnt = 0
    ws, ss = window_size, shift_size
    for d in (slice(-ws[0]), slice(-ws[0], -ss[0]), slice(-ss[0], None)):
        for h in (slice(-ws[1]), slice(-ws[1], -ss[1]), slice(-ss[1], None)):
            for w in (slice(-ws[2]), slice(-ws[2], -ss[2]), slice(-ss[2], None)):
                img[d, h, w] = cnt
                cnt += 1
    return img


def compute_mask(D, H, W, window_size, shift_size):
    """swin_transformer_3d.py:548-562.  -> float32 [nW, N, N] in {0, -100}."""
    img = region_ids(D, H, W, window_size, shift_size)[None, :, :, :, None].astype(np.float32)
    mw = window_partition(img, window_size)[..., 0]  # [nW, N]
    diff = mw[:, None, :] - mw[:, :, None]
    return np.where(diff != 0, np.float32(-100.0), np.float32(0.0))


def window_region_ids(D, H, W, window_size, shift_size):
    """Per-window token region ids [nW, N] (the compact form of compute_mask:
    mask[w,i,j] = -100 if rid[w,i] != rid[w,j] else 0)."""
    img = region_ids(D, H, W, window_size, shift_size)[None, :, :, :, None]
    return window_partition(img, window_size)[..., 0].astype(np.int32)


def shifted_window_token_index(D, H, W, window_size, shift_size):
    """Flat source index (into D*H*W) of every (window, token) after
    torch.roll(x, -shift) + window_partition (swin_transformer_3d.py:459-466).
    The reverse path (window_reverse + roll(+shift), :470-476) scatters through
    the same index.  -> int64 [nW, N]."""
    idx = np.arange(D * H * W, dtype=np.int64).reshape(1, D, H, W, 1)
    if any(s > 0 for s in shift_size):
        idx = np.roll(idx, shift=(-shift_size[0], -shift_size[1], -shift_size[2]), axis=(1, 2, 3))
    return window_partition(idx, window_size)[..., 0]


def mask_blend_weight(mask, T, H, W):
    """swin_transformer_3d.py:226-229.  mask [B,1,mh,mw] int -> w [B,1,T,H,W]
    (each mask cell repeated H//mh x W//mw spatially, broadcast over T)."""
    B, _, mh, mw = mask.shape
    w = mask[:, :, :, None, :, None]                       # B,1,mh,1,mw,1
    w = np.broadcast_to(w, (B, T, mh, H // mh, mw, W // mw))
    w = w.reshape(B, T, mh * (H // mh), mw * (W // mw))
    return w[:, None].copy()


def patch_merging_gather(x):
    """swin_transformer_3d.py:535-539.  x [B,D,H,W,C] -> [B,D,H/2,W/2,4C] in the
    reference's concat order (x0: even h even w, x1: odd h even w, x2: even h
    odd w, x3: odd h odd w); odd H/W are zero-padded first (:531-533)."""
    B, D, H, W, C = x.shape
    if H % 2 == 1 or W % 2 == 1:
        x = np.pad(x, ((0, 0), (0, 0), (0, H % 2), (0, W % 2), (0, 0)))
    x0 = x[:, :, 0::2, 0::2, :]
    x1 = x[:, :, 1::2, 0::2, :]
    x2 = x[:, :, 0::2, 1::2, :]
    x3 = x[:, :, 1::2, 1::2, :]
    return np.concatenate([x0, x1, x2, x3], -1)


def input_ssl_ids(token_ids, mlm_label):
    """multimodal_transformer_pretrain.py:97 — the un-masked caption."""
    return np.where(mlm_label == -100, token_ids, mlm_label)


def mlm_rows(mlm_label):
    """multimodal_transformer_pretrain.py:137-139 — flat row indices and labels
    of the masked positions."""
    flat = mlm_label.reshape(-1)
    idx = np.nonzero(flat != -100)[0]
    return idx.astype(np.int64), flat[idx]
