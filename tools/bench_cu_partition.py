#!/usr/bin/env python3
"""CU partitions (ABI 13): do kernels of two streams overlap when each stream owns a disjoint set of CUs?

Three workloads of one steady frame group at 270 x 480 (RefVSR_small):
  A  match_top2, 129 600 columns x 32 400 reference rows (P stream; 254 workgroups of 152 KB LDS: owns every CU it touches for ~1 ms)
  B  the backward branch's 24-block resblock24 chain as FOUR-map launches (M stream; 77 KB LDS, one workgroup per CU)
  C  the forward branch's 24-block chain, one map per launch (F stream; latency-bound launches of ~10 us)
Measured: each alone on the whole chip; A + B + C back to back on one stream (the serial sum); on three plain streams; on CU-masked
streams for a list of splits.  Wall = HIP events on the caller's stream around the fork / join."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from refvsr_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
h, w = 270, 480
C, n = 24, 24
raw = []
for _ in range(n):
    ws = [torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5 * 0.5 for _ in range(2)]
    raw.append(((ws[0], torch.zeros(C)), (ws[1], torch.zeros(C))))
ch = ops.Resblock24Chain(raw, dev)
xs4 = [ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev)) for _ in range(4)]
x1 = ops.pack_nhwc16(torch.randn(C, h, w, generator=g).to(dev))
lr_f = torch.randn(16, h, w, generator=g).to(dev)
ref_f = torch.randn(16, h // 2, w // 2, generator=g).to(dev)     # the engine matches against the 2x-pooled reference
lr_rows, inv_lr = ops.match_patches(lr_f, 512)
ref_rows, inv_ref = ops.match_patches(ref_f, 256)
npx = h * w
nref = (h // 2) * (w // 2)
REP = int(os.environ.get('REP', '4'))


def A():
    return ops.match_top2(ref_rows, nref, lr_rows, npx, 1)


def B():
    return ops.resblock24_chain_b(ch, xs4, 0.0)


def Cw():
    return ops.resblock24_chain(ch, x1, 0.0)


def wall(plan):
    """plan: list of (stream or None, [callables]); every stream's list is enqueued REP times, streams interleaved per repetition."""
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    keep = []
    e0.record()
    for st, _ in plan:
        if st is not None:
            st.wait_event(e0)
    for _ in range(REP):
        for st, fns in plan:
            with ops.on_stream(cur if st is None else st):
                for fn in fns:
                    keep.append(fn())
    for st, _ in plan:
        if st is not None:
            cur.wait_stream(st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP


def best(plan, k=5):
    for _ in range(2):
        wall(plan)
    return min(wall(plan) for _ in range(k))


def checks():
    idx, val = A()
    return int(idx.long().sum()), float(B().float().sum()), float(Cw().float().sum())


ref = checks()
tA, tB, tC = best([(None, [A])]), best([(None, [B])]), best([(None, [Cw])])
serial = best([(None, [A, B, Cw])])
print('whole chip, one stream: A match_top2 %.0f us | B 4-map chain %.0f us (%.2f us/launch) | C 1-map chain %.0f us (%.2f us/launch) | '
      'A+B+C back to back %.0f us (sum of parts %.0f)' % (tA, tB, tB / n, tC, tC / n, serial, tA + tB + tC), flush=True)
s3 = [torch.cuda.Stream() for _ in range(3)]
plain = best([(s3[0], [A]), (s3[1], [B]), (s3[2], [Cw])])
print('three plain streams: %.0f us = %.3f x serial' % (plain, plain / serial), flush=True)
total = ops.num_cus()
splits = [(128, 96, 32), (128, 64, 64), (160, 64, 32), (96, 128, 32), (144, 80, 32), (112, 112, 32), (128, 128, 0), (160, 96, 0), (192, 64, 0)]
for a, b, c in splits:
    if a + b + c != total:
        continue
    sa = ops.CuStream(0, a)
    sb = ops.CuStream(a, b)
    sc = ops.CuStream(a + b, c) if c else sb
    made = [sa, sb] + ([sc] if c else [])
    # each alone on its partition, then together
    pa, pb = best([(sa, [A])], 3), best([(sb, [B])], 3)
    pc = best([(sc, [Cw])], 3)
    plan = [(sa, [A]), (sb, [B, Cw])] if not c else [(sa, [A]), (sb, [B]), (sc, [Cw])]
    t = best(plan)
    with ops.on_stream(sa):
        idx, val = A()
    with ops.on_stream(sb):
        ob = B()
    with ops.on_stream(sc):
        oc = Cw()
    torch.cuda.synchronize()
    same = (int(idx.long().sum()), float(ob.float().sum()), float(oc.float().sum())) == ref
    print('CU split A %3d | B %3d | C %3d: alone on its share A %.0f  B %.0f  C %.0f us; together %.0f us = %.3f x serial (%.3f x plain streams); outputs %s'
          % (a, b, c, pa, pb, pc, t, t / serial, t / plain, 'identical' if same else 'DIFFER'), flush=True)
    for st in made:                                  # the budget table of the library holds 16 streams
        ops.hip.check(ops.hip.lib().refvsr_stream_set_cu_budget(ops.C.c_void_p(st.cuda_stream), 0), 'stream_set_cu_budget')
