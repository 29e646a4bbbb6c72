#!/usr/bin/env python3
"""Could the matching's VGG feature convs leave the fp32 matrix instructions (1/16 of the fp16 rate) without losing what the exact
arg-max needs?  CPU study: operands split in three fp16 terms  x = hi + mid * 2^-11 + lo * 2^-22
(mid, lo stored scaled, so they stay normal numbers), the six products  hi.hi | hi.mid + mid.hi | hi.lo + mid.mid + lo.hi  summed
in three accumulators -- each product of two fp16 values is exact in fp32 -- against

  (a) the exact result (float64),   (b) the float32 conv the reference runs (torch CPU),
  (c) the fp32-instruction path of this build modelled as float32 accumulation in another order.

Workload: VGG19's conv1_2 (64 -> 64, 3x3) on relu(conv1_1(image)) of a synthetic frame, and the 1x1 map 64 -> 16; what is reported:
relative error of every path against (a), and how many arg-max decisions of the cosine matching change against (a) when the
features carry that error (near-ties decide: the GPU build sees 1-3 flips of 129 600 against the reference today)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def split3(x):
    """x (float32 / float64) -> hi, mid, lo as float16 with  x ~ hi + mid * 2^-11 + lo * 2^-22."""
    x = x.astype(np.float64)
    hi = x.astype(np.float16)
    r1 = (x - hi.astype(np.float64)) * 2.0 ** 11
    mid = r1.astype(np.float16)
    r2 = (r1 - mid.astype(np.float64)) * 2.0 ** 11
    lo = r2.astype(np.float16)
    return hi, mid, lo


def conv_as_gemm(x, w):
    """x [C,H,W], w [O,C,3,3] -> columns [C*9, H*W], weights [O, C*9] (zero padding 1)."""
    C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    cols = np.stack([xp[:, dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], 1).reshape(C * 9, H * W)
    return cols, w.reshape(w.shape[0], -1)


def main():
    rng = np.random.default_rng(0)
    H, W = 64, 96
    img = rng.random((3, H, W)).astype(np.float32)
    w0 = (rng.standard_normal((64, 3, 3, 3)) * 0.3).astype(np.float32)
    w1 = (rng.standard_normal((64, 64, 3, 3)) * (2.0 / (64 * 9)) ** 0.5).astype(np.float32)
    c0, g0 = conv_as_gemm(img, w0)
    x1 = np.maximum(g0.astype(np.float64) @ c0.astype(np.float64), 0.0).astype(np.float32).reshape(64, H, W)   # relu(conv1_1), fp32 values
    cols, wm = conv_as_gemm(x1, w1)
    exact = wm.astype(np.float64) @ cols.astype(np.float64)
    ref32 = torch.nn.functional.conv2d(torch.from_numpy(x1)[None], torch.from_numpy(w1), padding=1)[0].numpy().reshape(64, -1)
    # (c) float32 accumulation in another order: K processed in chunks of 4 (the fp32 instruction's K), chunk sums added in order
    acc = np.zeros_like(exact, dtype=np.float32)
    K = cols.shape[0]
    for k0 in range(0, K, 4):
        acc = (acc + (wm[:, k0:k0 + 4].astype(np.float32) @ cols[k0:k0 + 4].astype(np.float32))).astype(np.float32)
    # split path: six products, three accumulators (float32 accumulation emulated per K chunk of 32 = one fp16 instruction)
    wh, wmid, wl = split3(wm)
    xh, xm, xl = split3(cols)
    a0 = np.zeros_like(acc)
    a1 = np.zeros_like(acc)
    a2 = np.zeros_like(acc)
    f = lambda a: a.astype(np.float32)
    for k0 in range(0, K, 32):
        s = slice(k0, k0 + 32)
        a0 = f(a0 + f(f(wh[:, s]) @ f(xh[s])))
        a1 = f(a1 + f(f(wh[:, s]) @ f(xm[s])) + f(f(wmid[:, s]) @ f(xh[s])))
        a2 = f(a2 + f(f(wh[:, s]) @ f(xl[s])) + f(f(wmid[:, s]) @ f(xm[s])) + f(f(wl[:, s]) @ f(xh[s])))
    split = f(f(a0 + a1 * np.float32(2.0 ** -11)) + a2 * np.float32(2.0 ** -22))
    # two-term split for comparison (what the conv kernels use for WEIGHTS: hi + lo, activations fp16)
    two = f(f(f(wh) @ f(xh)) + f(f(wmid) @ f(xh)) * np.float32(2.0 ** -11))
    scale = np.abs(exact).max()
    rel = lambda y: float(np.abs(y.astype(np.float64) - exact).max() / scale)
    rms = lambda y: float(np.sqrt(np.mean((y.astype(np.float64) - exact) ** 2)) / scale)
    print('VGG conv1_2 (64 -> 64, 3x3) on a %dx%d frame: max / rms error relative to the largest output, against float64' % (H, W))
    print('  float32 conv of the reference (torch CPU)            %.3e  %.3e' % (rel(ref32), rms(ref32)))
    print('  float32 accumulation, K in chunks of 4 (this build)  %.3e  %.3e' % (rel(acc), rms(acc)))
    print('  3 x fp16 split, six products, three accumulators     %.3e  %.3e' % (rel(split), rms(split)))
    print('  fp16 activations x (hi + lo) weights (the conv path) %.3e  %.3e' % (rel(two), rms(two)))
    print('  fp16 range: largest |activation| %.3g, smallest scaled lo term %.3g (fp16 normal range 6.1e-5 .. 65504)'
          % (float(np.abs(cols).max()), float(np.abs(xl.astype(np.float64))[np.abs(xl.astype(np.float64)) > 0].min())))
    # what the matching sees: cosine arg-max over patches of the mapped features (1x1 conv 64 -> 16, 3x3 patches), LR vs a shifted ref
    wmap = (rng.standard_normal((16, 64)) * 0.2).astype(np.float32)

    def argmax_of(feat64):
        fm = np.maximum(wmap.astype(np.float64) @ feat64, 0.2 * (wmap.astype(np.float64) @ feat64)).reshape(16, H, W)
        def patches(m):
            mp = np.pad(m, ((0, 0), (1, 1), (1, 1)), mode='reflect')
            p = np.stack([mp[:, dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], 1).reshape(144, H * W)
            return p / np.maximum(np.linalg.norm(p, axis=0, keepdims=True), 1e-12)
        lr = patches(fm)
        rf = patches(np.roll(fm, (2, 3), (1, 2))[:, ::2, ::2].repeat(2, 1).repeat(2, 2))
        return np.argmax(rf.T @ lr, 0)
    base = argmax_of(exact)
    for name, y in (('float32 conv of the reference', ref32), ('float32, chunks of 4', acc), ('3 x fp16 split', split), ('fp16 x (hi + lo)', two)):
        print('  arg-max decisions that change against float64 features, %-32s %d of %d' % (name + ':', int((argmax_of(y.astype(np.float64)) != base).sum()), base.size))


if __name__ == '__main__':
    main()
