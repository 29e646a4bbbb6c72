#!/usr/bin/env python3
"""Do two independent chains of LR ResBlock launches on two streams fill each other's launch boundaries?  24-block resblock24
chains at 270x480 (510 workgroups per launch = every CU holds its two workgroups): one chain alone, two chains back to back
on one stream, two chains on two streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
C, n = 24, 24
raw = []
for _ in range(n):
    ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5 for _ in range(2)]
    raw.append(((ws[0], torch.zeros(C)), (ws[1], torch.zeros(C))))
ch = ops.Resblock24Chain(raw, dev)
h, w = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '270x480').split('x'))
xa = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
xb = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(two_streams, both=True):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    if two_streams:
        sa.wait_event(e0)
        sb.wait_event(e0)
        for _ in range(4):                                   # enqueue alternately so that neither queue runs dry
            with ops.on_stream(sa):
                ops.resblock24_chain(ch, xa, 0.0)
            with ops.on_stream(sb):
                ops.resblock24_chain(ch, xb, 0.0)
        torch.cuda.current_stream().wait_stream(sa)
        torch.cuda.current_stream().wait_stream(sb)
    else:
        with ops.on_stream(torch.cuda.current_stream()):
            for _ in range(4):
                ops.resblock24_chain(ch, xa, 0.0)
            if both:
                for _ in range(4):
                    ops.resblock24_chain(ch, xb, 0.0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for _ in range(3):
    run(False), run(True)
one = min(run(False, False) for _ in range(5))
seq = min(run(False) for _ in range(5))
par = min(run(True) for _ in range(5))
print('%dx%d, %d-block chains x 4: one chain %.0f us (%.2f us/block); two chains on one stream %.0f us (%.2f us/block); two chains on two streams %.0f us (%.2f us/block) = %.2fx'
      % (h, w, n, one, one / (4 * n), seq, seq / (8 * n), par, par / (8 * n), seq / par))
