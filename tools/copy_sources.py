#!/usr/bin/env python3
"""Where do the device-to-device copy launches of a steady-state frame come from?  Runs the bench's frame loop under
torch.profiler (with stacks) and prints, per python call site inside refvsr_amd/ or bench.py, the number of aten::copy_ /
aten::clone / aten::cat / aten::zeros / aten::fill_ / aten::contiguous calls per frame and the GPU kernels they launch."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from refvsr_amd import SRNet, get_config, make_state_dict  # noqa: E402
from refvsr_amd.synth import make_clip, window_indices  # noqa: E402

dev = torch.device('cuda:0')
cfg = get_config('p', 'm', 'config_RefVSR_small_L1')
cfg.frame_num = 5
net = SRNet(cfg).to(dev).eval()
net.load_state_dict(make_state_dict(cfg, 1234))
net.Network.set_pipelined(True)
nfr = 16
lr, rf, gt = make_clip(nfr, 270, 480, seed=0)
lr, rf = lr.to(dev), rf.to(dev)
wins = [torch.tensor(window_indices(f, nfr, 5), device=dev) for f in range(nfr)]
lw = [lr[w][None].contiguous() for w in wins]
rw = [rf[w][None].contiguous() for w in wins]
torch.cuda.synchronize()


def run(f0, f1):
    for f in range(f0, f1):
        net(lw[f], rw[f], f == 0, frame_ids=window_indices(f, nfr, 5))


run(0, 6)
torch.cuda.synchronize()
NF = 8
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    run(6, 6 + NF)
    torch.cuda.synchronize()
OPS = ('aten::copy_', 'aten::clone', 'aten::cat', 'aten::zeros', 'aten::fill_', 'aten::contiguous', 'aten::zero_', 'aten::stack',
       'aten::to', 'aten::_to_copy', 'aten::empty_like', 'aten::max', 'aten::min', 'aten::sub', 'aten::div')
sites = collections.Counter()
for ev in prof.events():
    if ev.name in OPS and ev.stack:
        site = next((s for s in ev.stack if 'refvsr_amd/' in s or 'bench.py' in s or 'copy_sources' in s), None)
        if site is None:
            continue
        site = site.split('/root/repo/')[-1].split(os.path.basename(ROOT) + '/')[-1]
        sites[(ev.name, site)] += 1
print('per frame (%d profiled frames):' % NF)
for (name, site), n in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print('  %6.2f  %-18s %s' % (n / NF, name, site))
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith('CUDA') and ('copy' in ev.name.lower() or 'fill' in ev.name.lower() or 'Memcpy' in ev.name or 'Memset' in ev.name or 'at::native' in ev.name):
        kern[ev.name[:90]] += 1
print('GPU-side copy / fill / ATen kernels per frame:')
for k, n in kern.most_common(20):
    print('  %6.2f  %s' % (n / NF, k))
