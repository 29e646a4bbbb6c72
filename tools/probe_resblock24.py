#!/usr/bin/env python3
"""Where does the 24-channel fused ResBlock kernel (csrc/resblock24.hip) spend its time?  s_memtime stamps recorded by wave 0
of every workgroup (refvsr_set_probe switches the launches to the PROBE instantiation) at twelve points of the kernel, for the
LR (510 workgroups, one tile each), LR/2 (136) and 2x (512 persistent workgroups, 4 tiles each) maps.  Cycles."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from refvsr_amd import hip, ops  # noqa: E402

dev = torch.device('cuda:0')
STAGES = ['entry -> prologue loads issued', 'loads issued -> first barrier passed', '(tile start) -> conv1 K loop done',
          'fold + residual reads + barrier A', 'epilogue 1 (t -> LDS) + prefetch issue', 'barrier B', 'conv2 K loop',
          'barrier C + park next tile', 'fold + stores issued', 'barrier D']


def main():
    g = torch.Generator().manual_seed(0)
    Cc = 24
    ws = [torch.randn(Cc, Cc, 3, 3, generator=g) / (Cc * 9) ** 0.5 * 0.5 for _ in range(2)]
    ch = ops.Resblock24Chain([((ws[0], torch.zeros(Cc)), (ws[1], torch.zeros(Cc)))], dev)
    probe = torch.zeros(512 * 12, dtype=torch.int64, device=dev)
    if os.environ.get('PROBE_WAVES'):                      # 8: 8 x 32 tiles; 16: 16 x 32 tiles on sixteen waves; default: by map size
        hip.lib().refvsr_set_resblock24_waves(int(os.environ['PROBE_WAVES']))
    for name, h, w, it in (('LR', 270, 480, 0), ('LR/2', 135, 240, 0), ('2x first tile', 540, 960, 0), ('2x 2nd tile', 540, 960, 1),
                           ('2x 4th tile', 540, 960, 3)):
        x = ops.pack_nhwc16(torch.randn(Cc, h, w, generator=g).to(dev))
        y = x
        for _ in range(5):
            y = ops.resblock24_chain(ch, y, 0.0)
        torch.cuda.synchronize()
        hip.lib().refvsr_set_probe(C.c_void_p(probe.data_ptr()), it)
        reps = []
        for rep in range(6):
            probe.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = ops.resblock24_chain(ch, y, 0.0)
            e1.record()
            torch.cuda.synchronize()
            pa = probe.view(512, 12).cpu()
            keep = (pa[:, 0] > 0) & (pa[:, 9] > 0)
            p = pa[keep].double()
            # (s_memtime counters are not common to the chip -- per XCD / clock domain -- so only differences inside one
            #  workgroup are meaningful: no entry skew / first-entry-to-last-exit figures)
            reps.append((e0.elapsed_time(e1) * 1e3, p))
        hip.lib().refvsr_set_probe(None, 0)
        ev_us, p = reps[-1]
        print('== %s (%dx%d, tile iteration %d): %d workgroups stamped, event time %.1f us' % (name, h, w, it, p.shape[0], ev_us))
        d = p[:, 1:11] - p[:, 0:10]
        for i, s in enumerate(STAGES):
            if it > 0 and i < 3:
                continue
            print('   %-42s %8.0f %8.0f %8.0f' % (s, float(d[:, i].mean()), float(d[:, i].min()), float(d[:, i].max())))
        print('   tile: conv1 K loop done -> barrier D        %8.0f;   workgroup entry -> exit %8.0f' %
              (float((p[:, 10] - p[:, 3]).mean()), float((p[:, 11] - p[:, 0]).mean())))
        print('   event times of the 6 repeats (us): ' + ' '.join('%.1f' % r[0] for r in reps))


if __name__ == '__main__':
    main()
