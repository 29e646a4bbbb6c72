#!/usr/bin/env python3
"""Where does the 48-channel fused ResBlock kernel (csrc/resblock48.hip) spend its time?  s_memtime stamps recorded by wave 0 of
every workgroup (refvsr_set_probe switches the ReLU launches to the PROBE instantiation) at twelve points of the kernel, for the
LR map of RefVSR_MFID (270 x 480: 510 tiles of 8 x 32 on 256 persistent workgroups, two tiles each), LR/2 and the 2x map.  Shader cycles
(s_memtime), differences inside one workgroup only; one v_mfma_f32_16x16x32_f16 occupies a SIMD's matrix pipe for 16 of them.

VERDICT r3 item 6: "build the PROBE variant and show cycles" -- the output goes to profiles/r04_resblock48_probe.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402

from refvsr_amd import hip, ops  # noqa: E402

dev = torch.device('cuda:0')
STAGES = ['entry -> start of the probed tile (tile 0: W1 84 KB + biases + x tile landed)', '(tile start) -> conv1 K loop done (22 groups x 14 steps x 6)',
          'residual reads + next-tile fetch issue + barrier A', 'W2 DMA issue + t -> LDS (asm stores) drained', 'barrier B (W2 landed)',
          'conv2 K loop (16 groups x 14 steps x 6)', 'barrier C', 'W1 DMA issue + park next x tile', 'epilogue + output stores issued',
          'barrier D (W1 landed)']
# MFMA issue cycles of one tile on one CU (four SIMDs): (22 + 16) groups x 14 K-steps x 6 fragments x 16 cycles / 4 SIMDs
MFMA_TILE = (22 + 16) * 14 * 6 * 16 // 4


def main():
    g = torch.Generator().manual_seed(0)
    Cc = 48
    ws = [torch.randn(Cc, Cc, 3, 3, generator=g) / (Cc * 9) ** 0.5 * 0.5 for _ in range(2)]
    ch = ops.Resblock48Chain([((ws[0], torch.zeros(Cc)), (ws[1], torch.zeros(Cc)))], dev)
    probe = torch.zeros(512 * 12, dtype=torch.int64, device=dev)
    print('MFMA issue of one 8 x 32 tile on one CU: %d cycles (conv1 on the 10 x 34 halo region = 22 groups, conv2 = 16 groups; '
          '14 K-steps x 6 fragments x 16 cycles, four SIMDs)' % MFMA_TILE)
    for name, h, w, it in (('LR first tile', 270, 480, 0), ('LR second tile', 270, 480, 1), ('LR/2 (one tile per workgroup)', 135, 240, 0),
                           ('2x second tile', 540, 960, 1), ('2x eighth (last) tile', 540, 960, 7)):
        x = ops.pack_nhwc16(torch.randn(Cc, h, w, generator=g).to(dev))
        y = x
        for _ in range(5):
            y = ops.resblock48_chain(ch, y, 0.0)
        torch.cuda.synchronize()
        hip.lib().refvsr_set_probe(C.c_void_p(probe.data_ptr()), it)
        reps = []
        for rep in range(6):
            probe.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            y = ops.resblock48_chain(ch, y, 0.0)
            e1.record()
            torch.cuda.synchronize()
            pa = probe.view(512, 12).cpu()
            keep = (pa[:, 0] > 0) & (pa[:, 9] > 0)
            reps.append((e0.elapsed_time(e1) * 1e3, pa[keep].double()))
        hip.lib().refvsr_set_probe(None, 0)
        ev_us, p = reps[-1]
        print('== %s (%dx%d, tile iteration %d): %d workgroups stamped, event time of the probed launch %.1f us' % (name, h, w, it, p.shape[0], ev_us))
        if p.shape[0] == 0:
            continue
        d = p[:, 1:11] - p[:, 0:10]
        print('   %-62s %8s %8s %8s' % ('stage (cycles)', 'mean', 'min', 'max'))
        for i, s in enumerate(STAGES):
            print('   %-62s %8.0f %8.0f %8.0f' % (s, float(d[:, i].mean()), float(d[:, i].min()), float(d[:, i].max())))
        tile = p[:, 10] - p[:, 1]
        print('   tile start -> barrier D  %8.0f cycles   (MFMA issue %d = %.0f %%);   workgroup entry -> exit %8.0f' %
              (float(tile.mean()), MFMA_TILE, 100.0 * MFMA_TILE / float(tile.mean()), float((p[:, 11] - p[:, 0]).mean())))
        print('   event times of the 6 repeats (us): ' + ' '.join('%.1f' % r[0] for r in reps))


if __name__ == '__main__':
    main()
