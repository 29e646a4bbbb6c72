#!/usr/bin/env python3
"""Where does a single-tile-per-CU fused ResBlock launch spend its time?  s_memtime stamps recorded by every workgroup
(refvsr_set_probe) at eight points of the kernel, for the LR (255 workgroups, one tile each), LR/2 (72) and 2x
(256 persistent workgroups, 8 tiles each; first tile stamped) maps.  Prints stage durations in ns."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from refvsr_amd import hip, ops  # noqa: E402
from refvsr_amd.packing import pack_conv  # noqa: E402

dev = torch.device('cuda:0')
STAGES = ['entry->loads issued', 'loads issued->landed (sync)', 'conv1 K loop', 'conv1 epilogue (t -> LDS)', 'barrier',
          'conv2 K loop', 'conv2 epilogue (stores issued)']


def main():
    g = torch.Generator().manual_seed(0)
    Cc = 24
    w1 = torch.randn(Cc, Cc, 3, 3, generator=g) / (Cc * 9) ** 0.5
    c1 = ops.ConvWeights(pack_conv(w1, torch.zeros(Cc), [Cc]), dev)
    c2 = ops.ConvWeights(pack_conv(w1.flip(0), torch.zeros(Cc), [Cc]), dev)
    probe = torch.zeros(512 * 12, dtype=torch.int64, device=dev)     # one slot set per workgroup (the lean kernel launches up to 512)
    for name, h, w, it in (('LR', 270, 480, 0), ('LR/2', 135, 240, 0), ('2x first tile', 540, 960, 0), ('2x 4th tile', 540, 960, 3)):
        x = ops.pack_nhwc16(torch.randn(Cc, h, w, generator=g).to(dev))
        y = x
        for _ in range(5):
            y = ops.resblock(c1, c2, y, act=0.0)
        torch.cuda.synchronize()
        hip.lib().refvsr_set_probe(C.c_void_p(probe.data_ptr()), it)
        reps = []
        for rep in range(6):
            probe.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = ops.resblock(c1, c2, y, act=0.0)
            e1.record()
            torch.cuda.synchronize()
            p = probe.view(512, 12).cpu()
            p = p[(p[:, 0] > 0) & (p[:, 7] > 0)].double()
            reps.append((e0.elapsed_time(e1) * 1e3, p))
        hip.lib().refvsr_set_probe(None, 0)
        ev_us, p = reps[-1]
        # s_memtime ticks: calibrate against the constant 100 MHz guess by comparing span with the event time
        span = float(p[:, 10].max() - p[:, 0].min())
        rt = float((p[:, 9] - p[:, 8]).mean())          # 100 MHz ticks entry -> exit
        cyc = float((p[:, 10] - p[:, 0]).mean())
        print('   shader clock during the kernel: %.0f cycles in %.2f us -> %.2f GHz' % (cyc, rt / 100.0, cyc / (rt * 10.0) if rt > 0 else 0))
        d = p[:, 1:8] - p[:, :7]
        print('== %s (%dx%d): %d workgroups stamped, event time %.1f us, in-kernel span %.0f ticks' % (name, h, w, p.shape[0], ev_us, span))
        print('   first entry -> last entry skew: %.0f ticks; per-stage ticks (mean / min / max over workgroups):' %
              float(p[:, 0].max() - p[:, 0].min()))
        for i, s in enumerate(STAGES):
            if it > 0 and i < 2:
                continue
            print('   %-34s %8.0f %8.0f %8.0f' % (s, float(d[:, i].mean()), float(d[:, i].min()), float(d[:, i].max())))
        print('   total entry -> stores issued        %8.0f %8.0f %8.0f' % (float((p[:, 7] - p[:, 0]).mean()), float((p[:, 7] - p[:, 0]).min()),
                                                                            float((p[:, 7] - p[:, 0]).max())))
        print('   event times of the 6 repeats (us): ' + ' '.join('%.1f' % r[0] for r in reps))


if __name__ == '__main__':
    main()
