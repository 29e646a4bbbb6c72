#!/usr/bin/env python3
"""Workgroup shapes of the fused 24-channel block (refvsr_set_resblock24_waves): 8 waves on 8 x 32 tiles (two workgroups per CU),
16 waves on 16 x 32 (one per CU; the default of the large maps) and -- round 6 -- 8 waves on 16 x 32 tiles with split passes (816: two
workgroups per CU).  Device microseconds per launch of a 24-block chain queued behind a blocker (host out of the picture), for the
single-map LR launch, the four-map LR launch and the single 2x launch; outputs compared bit for bit with the default shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import hip, ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
C, n = 24, 24
raw = []
for _ in range(n):
    ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5 for _ in range(2)]
    raw.append(((ws[0], torch.randn(C, generator=g) * 0.05), (ws[1], torch.randn(C, generator=g) * 0.05)))
ch = ops.Resblock24Chain(raw, dev)
lib = hip.lib()


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    blocker = torch.randn(8192, 8192, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _ = blocker @ blocker
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3 / n


cases = [('LR 270x480 B=1', 270, 480, 1), ('LR 270x480 B=4', 270, 480, 4), ('2x 540x960 B=1', 540, 960, 1), ('2x 540x960 B=4', 540, 960, 4),
         ('odd 67x101 B=3', 67, 101, 3)]
for name, h, w, B in cases:
    xs = [ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev)) for _ in range(B)]
    for act in (0.0, 0.2):
        fn = (lambda: ops.resblock24_chain(ch, xs[0], act)) if B == 1 else (lambda: ops.resblock24_chain_b(ch, xs, act))
        lib.refvsr_set_resblock24_waves(0)
        want = fn()
        want = want.clone() if torch.is_tensor(want) else torch.stack(list(want)).clone()
        row = []
        for wv in (8, 16, 816):
            if lib.refvsr_set_resblock24_waves(wv) != 0:          # (816: the split-pass shape of round 6 -- measured, not shipped)
                continue
            got = fn()
            got = got if torch.is_tensor(got) else torch.stack(list(got))
            same = torch.equal(got, want)
            row.append('%s %6.2f us%s' % (wv, timeit(fn), '' if same else ' DIFFERS'))
        lib.refvsr_set_resblock24_waves(0)
        print('%-16s act %.1f: %s' % (name, act, ' | '.join(row)), flush=True)
