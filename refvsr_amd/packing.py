"""Host-side packing of conv weights into the MFMA fragment order consumed by
refvsr_conv_mfma (refvsr_amd/csrc/conv_mfma.hip).

GEMM view: D[row][pixel] = sum_k Wk[row][k] * X[k][pixel].
  k-block g = tap*ncg + cg (tap = ky*ks + kx, cg = 8-channel group of the concatenated, padded
  input), 8 halfs per block; K-step s covers blocks 4s..4s+3; lane l of a wave supplies block
  4s + (l>>4) for row (l&15) of a 16-row tile.
  packed[z][s][m][lane][8] = Wk[(z*MT+m)*16 + (lane&15)][(4s + (lane>>4))*8 : +8]
Rows are the conv output channels, except for pixel-shuffle convs where row r = sub*C + c holds
conv channel c*4 + sub so that a lane's 4 consecutive rows are 4 consecutive channels of ONE
output pixel (mmedit upsample.py:49-50 pixel_shuffle fused into the store).
"""
import numpy as np
import torch


def _pad8(c):
    return (c + 7) // 8 * 8


def choose_mt(cout):
    n_mt = (cout + 15) // 16
    if n_mt <= 3:
        return n_mt
    if n_mt % 3 == 0:
        return 3
    if n_mt % 2 == 0:
        return 2
    return 3


def kmatrix(w, src_channels, shuffle=False):
    """Wk [rows, G*8] (fp32 numpy) + row->conv-channel map.  src_channels: real channel count of each
    concatenated nhwc16 source (each is padded to a multiple of 8 in its own buffer)."""
    w = np.asarray(w, np.float32)
    cout, cin, ks, _ = w.shape
    assert sum(src_channels) == cin, (src_channels, cin)
    pads = [_pad8(c) for c in src_channels]
    ncg = sum(pads) // 8
    # padded-channel index -> original input channel (or -1)
    cmap = []
    base = 0
    for c, p in zip(src_channels, pads):
        cmap += [base + i if i < c else -1 for i in range(p)]
        base += c
    cmap = np.asarray(cmap)
    rows = np.arange(cout)
    if shuffle:
        C = cout // 4
        rows = (np.arange(cout) % C) * 4 + np.arange(cout) // C      # row r -> conv channel
    G = ks * ks * ncg
    Wk = np.zeros((cout, G * 8), np.float32)
    valid = cmap >= 0
    for tap in range(ks * ks):
        ky, kx = divmod(tap, ks)
        blk = np.zeros((cout, ncg * 8), np.float32)
        blk[:, valid] = w[rows][:, cmap[valid], ky, kx]
        Wk[:, tap * ncg * 8:(tap + 1) * ncg * 8] = blk
    return Wk, rows, ncg


def pack_conv(w, b, src_channels, shuffle=False, mt=None):
    """Returns dict(wpack fp16 [nz,S,MT,64,8], bias fp32 [nz*MT*16], cout, ksteps, mt, ksize, cpads)."""
    w = w.detach().cpu().float().numpy() if isinstance(w, torch.Tensor) else np.asarray(w, np.float32)
    b = b.detach().cpu().float().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float32)
    cout, cin, ks, _ = w.shape
    Wk, rows, ncg = kmatrix(w, src_channels, shuffle)
    G = ks * ks * ncg
    S = (G + 3) // 4
    MT = mt or choose_mt(cout)
    n_mt = (cout + 15) // 16
    nz = (n_mt + MT - 1) // MT
    R = nz * MT * 16
    full = np.zeros((R, S * 32), np.float32)
    full[:cout, :G * 8] = Wk
    # [nz, MT, lr, S, q, 8] -> [nz, S, MT, q, lr, 8]
    frag = full.reshape(nz, MT, 16, S, 4, 8).transpose(0, 3, 1, 4, 2, 5).reshape(nz, S, MT, 64, 8)
    bias = np.zeros(R, np.float32)
    bias[:cout] = b[rows]
    return dict(wpack=torch.from_numpy(np.ascontiguousarray(frag)).to(torch.float16),
                bias=torch.from_numpy(bias), cout=cout, ksteps=S, mt=MT, ksize=ks,
                cpads=[_pad8(c) for c in src_channels], shuffle=shuffle)
