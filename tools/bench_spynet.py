#!/usr/bin/env python3
"""SPyNet on the MI355X kernels at the BASELINE size: time of one flow (270x480 pair, six pyramid levels, 31 launches) and its
deviation from (a) the reference fixture (tests/golden/op_spynet.npz, small size) and (b) the hi + lo flow at full size.
Knobs: REFVSR_SPYNET_HILO=1 (hi + lo weights in the streamed 7x7 convs), REFVSR_CONV_NO_NW8=1 (4 waves x 4 pixel groups)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from refvsr_amd import get_config, make_state_dict  # noqa: E402
from refvsr_amd.engine import Engine, FrameCtx, Weights  # noqa: E402
from refvsr_amd.synth import make_clip  # noqa: E402

dev = torch.device('cuda:0')
cfg = get_config('p', 'm', 'config_RefVSR_small_L1')
cfg.frame_num = 5
sd = make_state_dict(cfg, 1234)
eng = Engine(cfg, Weights(cfg, sd, dev))
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'op_spynet.npz'))
a, b = torch.from_numpy(g['a'])[0].to(dev), torch.from_numpy(g['b'])[0].to(dev)
fl = eng.flow(FrameCtx(a, a), FrameCtx(b, b)).cpu()
err = float((fl - torch.from_numpy(g['flow'])[0]).abs().max())
lr, rf, gt = make_clip(2, 270, 480, seed=0)
fa, fb = FrameCtx(lr[0].to(dev), lr[0].to(dev)), FrameCtx(lr[1].to(dev), lr[1].to(dev))
full = eng.flow(fa, fb)
torch.cuda.synchronize()
ts = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(5):
    e0.record()
    for _ in range(10):
        eng.flow_cache.clear()
        eng.flow(fa, fb)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
out = os.environ.get('SPYNET_DUMP')
if out:
    torch.save(full.cpu(), out)
cmp_ = os.environ.get('SPYNET_CMP')
d = float((full.cpu() - torch.load(cmp_)).abs().max()) if cmp_ and os.path.exists(cmp_) else float('nan')
print('spynet flow 270x480 [%s%s]: %.1f us per flow (min of 5x10; all %s) | vs reference fixture %.2e px | vs dumped flow %.2e px (|flow| max %.2f)'
      % ('hi+lo' if os.environ.get('REFVSR_SPYNET_HILO') == '1' else 'hi only', ', no nw8' if os.environ.get('REFVSR_CONV_NO_NW8') else '',
         min(ts), ' '.join('%.0f' % t for t in ts), err, d, float(full.abs().max())))
# round 4: the two flows a frame needs as ONE batched pass (Engine.flows: RefvsrConv.batch, refvsr_spynet_level_input_batch)
fc = FrameCtx(lr[1].to(dev), lr[1].to(dev))
pairs = [(fa, fb), (fb, fa)]
eng.flow_cache.clear()
both = eng.flows(pairs)
torch.cuda.synchronize()
eng.flow_cache.clear()
single = [eng.flow(x, y) for x, y in pairs]
same = all(torch.equal(p, q) for p, q in zip(both, single))
ts2 = []
for _ in range(5):
    e0.record()
    for _ in range(10):
        eng.flow_cache.clear()
        eng.flows(pairs)
    e1.record()
    torch.cuda.synchronize()
    ts2.append(e0.elapsed_time(e1) / 10 * 1e3)
print('spynet two flows 270x480 batched: %.1f us per PAIR of flows (min of 5x10; all %s) = %.1f us per flow; equal to the single passes: %s'
      % (min(ts2), ' '.join('%.0f' % t for t in ts2), min(ts2) / 2, same))
