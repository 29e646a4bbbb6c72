"""Deterministic synthetic RealMCVSR-like clips (no dataset is available offline).

HR ground truth = band-limited moving texture; LR (ultra-wide) = 4x area downsample; Ref (wide) =
central half-FoV crop of the GT downsampled 2x to the LR size -- the UW/W geometry of
/root/reference/data_loader/utils.py:55-60 -- all quantised to 8 bit like read_frame (:20,28).
"""
import numpy as np
import torch


def make_clip(nframes, h, w, seed=0, scale=4, start=0, want_gt=True):
    """Returns (lr [T,3,h,w], ref [T,3,h,w], gt [T,3,scale*h,scale*w]) float32 in [0,1] (CPU).
    Frame k of the returned clip is global frame `start + k` of an endless clip, so shards of one
    long clip can be generated independently.  want_gt=False: gt is None (the HR clip is 16x the LR clip: 25 MB per
    270x480 frame -- callers that only need inputs must not keep it)."""
    rs = np.random.RandomState(seed)
    H, W = h * scale, w * scale
    ncomp = 10
    fy = rs.uniform(0.01, 0.45, (3, ncomp)).astype(np.float32)
    fx = rs.uniform(0.01, 0.45, (3, ncomp)).astype(np.float32)
    ph = rs.uniform(0, 2 * np.pi, (3, ncomp)).astype(np.float32)
    am = rs.uniform(0.2, 1.0, (3, ncomp)).astype(np.float32)
    noise = rs.uniform(-1, 1, (3, H // 4 + 64, W // 4 + 64)).astype(np.float32)
    vx, vy = 2.0, 1.0                                  # HR pixels per frame
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    lrs, refs, gts = [], [], []
    for k in range(nframes):
        f = start + k
        oy, ox = vy * f, vx * f
        img = np.zeros((3, H, W), np.float32)
        for c in range(3):
            for j in range(ncomp):
                img[c] += am[c, j] * np.sin(fy[c, j] * (yy + oy) + fx[c, j] * (xx + ox) + ph[c, j])
        img = img / (2.0 * np.sqrt(ncomp)) + 0.5
        ny = (np.arange(H) // 4 + int(oy) // 4) % noise.shape[1]
        nx = (np.arange(W) // 4 + int(ox) // 4) % noise.shape[2]
        img += 0.06 * noise[:, ny][:, :, nx]
        gt = np.clip(img, 0.05, 0.95)
        lr = gt.reshape(3, h, scale, w, scale).mean((2, 4))
        y0, x0 = H // 4, W // 4
        crop = gt[:, y0:y0 + H // 2, x0:x0 + W // 2]
        ref = crop.reshape(3, h, 2, w, 2).mean((2, 4))
        q = lambda a: np.round(a * 255.0) / 255.0
        lrs.append(q(lr))
        refs.append(q(ref))
        if want_gt:
            gts.append(q(gt))
    to = lambda a: torch.from_numpy(np.stack(a).astype(np.float32))
    return to(lrs), to(refs), (to(gts) if want_gt else None)


def window_indices(f, nframes, t):
    """Frame indices of the sliding window centred on output frame f, edge frames repeated
    (data_loader/datasets.py:222-234)."""
    return [min(max(f - t // 2 + k, 0), nframes - 1) for k in range(t)]
