#!/usr/bin/env python3
"""GPU micro-benchmarks of the hot kernels (run on the MI355X box): match_top2 schedule variants
(A/B inside one process, results must be identical) and representative conv shapes.
Prints one line per case; also written to gpurun_out/bench_kernels.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

OUT = os.path.join(ROOT, 'gpurun_out', 'bench_kernels.txt')
dev = torch.device('cuda:0')


def emit(line):
    print(line, flush=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, 'a') as f:
        f.write(line + '\n')


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def bench_match():
    g = torch.Generator().manual_seed(0)
    h, w = 270, 480
    lr_f = torch.randn(16, h, w, generator=g).to(dev)
    ref_f = torch.randn(16, h // 2, w // 2, generator=g).to(dev)
    lr_rows, inv_lr = ops.match_patches(lr_f, 512)
    ref_rows, inv_ref = ops.match_patches(ref_f, 256)
    n_lr, n_ref = h * w, (h // 2) * (w // 2)
    flops = 2.0 * n_lr * n_ref * 144
    base = ops.match_top2(ref_rows, n_ref, lr_rows, n_lr, 1)
    us = timeit(lambda: ops.match_top2(ref_rows, n_ref, lr_rows, n_lr, 1), iters=10)
    emit('match_top2: %.1f us  %.1f TFLOP/s  (%.1f%% of 2.5 PF)' % (us, flops / us / 1e6, flops / us / 1e6 / 25.0))
    _, _, ref_lo = ops.match_patches(ref_f, 256, want_lo=True)
    _, _, lr_lo = ops.match_patches(lr_f, 512, want_lo=True)
    split = ((lr_rows, lr_lo), (ref_rows, ref_lo))
    us = timeit(lambda: ops.match_refine(lr_f, ref_f, inv_lr, inv_ref, base[0]), iters=10)
    emit('match_refine (re-rank only): %.1f us' % us)
    for margin in (ops.MATCH_EXACT_MARGIN, 5e-3, 2e-2):
        fl = ops.match_refine(lr_f, ref_f, inv_lr, inv_ref, base[0], base[1], margin, *split)[2]
        us = timeit(lambda: ops.match_refine(lr_f, ref_f, inv_lr, inv_ref, base[0], base[1], margin, *split), iters=10)
        emit('match_refine + exact search, margin %.1e: %d of %d columns flagged, %.1f us' % (margin, int(fl[0]), n_lr, us))
    us = timeit(lambda: ops.match_patches(lr_f, 512), iters=10)
    emit('match_patches(lr): %.1f us' % us)
    us = timeit(lambda: ops.match_patches(lr_f, 512, want_lo=True), iters=10)
    emit('match_patches(lr) with lo rows: %.1f us' % us)


# NOTE: timeit() launches back to back from Python; anything below ~12 us per call is bounded by the host launch rate,
# not by the kernel.  For the short kernels run this script under rocprofv3 (tools/gpu_runs/microbench_rocprof.sh) and
# read the per-kernel device durations.
def bench_conv():
    g = torch.Generator().manual_seed(1)
    cases = [  # name, cout, cins, ks, stride, h, w, shuffle, f32
        ('LR 24->24 3x3', 24, [24], 3, 1, 270, 480, False, False),
        ('LR 3+24->24 3x3', 24, [3, 24], 3, 1, 270, 480, False, False),
        ('LR 24+24->24 3x3', 24, [24, 24], 3, 1, 270, 480, False, False),
        ('LR 24->96 3x3 shuffle', 96, [24], 3, 1, 270, 480, True, False),
        ('2x 24->24 3x3', 24, [24], 3, 1, 540, 960, False, False),
        ('2x 24+24->24 3x3', 24, [24, 24], 3, 1, 540, 960, False, False),
        ('2x 32->32 3x3', 32, [32], 3, 1, 540, 960, False, False),
        ('2x 3->32 5x5', 32, [3], 5, 1, 540, 960, False, False),
        ('2x 32+32->32 5x5 s2', 32, [32, 32], 5, 2, 540, 960, False, False),
        ('2x 24->96 3x3 shuffle', 96, [24], 3, 1, 540, 960, True, False),
        ('HR 24->24 3x3', 24, [24], 3, 1, 1080, 1920, False, False),
        ('HR 24->3 3x3', 3, [24], 3, 1, 1080, 1920, False, False),
        ('spynet 288x480 32->64 7x7', 64, [32], 7, 1, 288, 480, False, False),
        ('spynet 288x480 64->32 7x7', 32, [64], 7, 1, 288, 480, False, False),
        ('spynet 288x480 8->32 7x7', 32, [8], 7, 1, 288, 480, False, False),
        ('vgg f32 3->64 3x3', 64, [3], 3, 1, 270, 480, False, True),
        ('vgg f32 64->64 3x3', 64, [64], 3, 1, 270, 480, False, True),
        ('vgg f32 64->16 1x1', 16, [64], 1, 1, 270, 480, False, True),
    ]
    for name, co, cins, ks, st, h, w, shuf, f32 in cases:
        cin = sum(cins)
        wt = torch.randn(co, cin, ks, ks, generator=g) / (cin * ks * ks) ** 0.5
        b = torch.zeros(co)
        cw = ops.ConvWeights(pack_conv(wt, b, cins, shuf, f32=f32), dev)
        srcs = []
        for c in cins:
            x = torch.randn(c, h, w, generator=g).to(dev)
            srcs.append(ops.pack_nhwc32(x) if f32 else ops.pack_nhwc16(x))
        fn = lambda: ops.conv(cw, srcs[0], srcs[1] if len(srcs) > 1 else None, stride=st, act=0.2, planar_out=(co % 4 != 0))
        us = timeit(fn)
        ho, wo = (h + st - 1) // st, (w + st - 1) // st
        fl = 2.0 * ho * wo * co * cin * ks * ks
        emit('conv %-28s %8.1f us  %7.1f TFLOP/s (useful)' % (name, us, fl / us / 1e6))
    # fused residual block vs two launches
    for name, h, w in (('LR', 270, 480), ('LR/2', 135, 240), ('2x', 540, 960)):
        C = 24
        w1 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
        c1 = ops.ConvWeights(pack_conv(w1, torch.zeros(C), [C]), dev)
        c2 = ops.ConvWeights(pack_conv(w1.flip(0), torch.zeros(C), [C]), dev)
        x = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
        us_f = timeit(lambda: ops.resblock(c1, c2, x, act=0.0))
        us_t = timeit(lambda: ops.conv(c2, ops.conv(c1, x, act=0.0), res=x))
        fl = 2 * 2.0 * h * w * C * C * 9
        emit('resblock %-5s fused %7.1f us (%6.1f TFLOP/s useful)   two launches %7.1f us' % (name, us_f, fl / us_f / 1e6, us_t))
    # launch floor: smallest possible conv
    wt = torch.randn(24, 24, 3, 3) * 0.1
    cw = ops.ConvWeights(pack_conv(wt, torch.zeros(24), [24]), dev)
    x = ops.pack_nhwc16(torch.randn(24, 8, 32).to(dev))
    emit('conv launch floor (8x32 px): %.1f us' % timeit(lambda: ops.conv(cw, x), iters=200))
    y = torch.randn(3, 270, 480).to(dev)
    emit('pack_nhwc16 270x480 (python+launch floor): %.1f us' % timeit(lambda: ops.pack_nhwc16(y, 8), iters=200))


if __name__ == '__main__':
    if os.path.exists(OUT):
        os.remove(OUT)
    bench_match()
    bench_conv()
