import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from refvsr_amd import ops
from refvsr_amd.packing import pack_conv
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
w = torch.randn(32, 64, 5, 5, generator=g) * 0.03
b = torch.randn(32, generator=g) * 0.1
for (h, wd) in ((61, 93), (135, 240), (270, 480), (540, 960)):
    r = ops.pack_nhwc16(torch.randn(32, h, wd, generator=g).to(dev))
    q = ops.pack_nhwc16(torch.randn(32, h, wd, generator=g).to(dev))
    for mt in (2, 1):
        cw = ops.ConvWeights(pack_conv(w, b, [32, 32], mt=mt), dev)
        for _ in range(3):
            ops.conv(cw, r, q, stride=2, act=0.2)
        torch.cuda.synchronize()
        evs = []
        for i in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv(cw, r, q, stride=2, act=0.2); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(50):
            ops.conv(cw, r, q, stride=2, act=0.2)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 50 * 1e6
        print('%dx%d mt=%d per-launch us: %s   wall per launch (50 back to back) %.1f us' % (h, wd, mt, ' '.join('%.0f' % (a.elapsed_time(b_) * 1e3) for a, b_ in evs), wall), flush=True)
