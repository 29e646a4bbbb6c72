#!/usr/bin/env python3
"""Which torch ops issue device copies / fills in a steady-state call?  (torch profiler, 6 pipelined calls)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from refvsr_amd import SRNet, get_config, make_state_dict
from refvsr_amd.synth import make_clip, window_indices
dev = torch.device('cuda:0')
cfg = get_config('b', 'b', 'config_RefVSR_small_L1'); cfg.frame_num = 5
net = SRNet(cfg).to(dev).eval(); net.load_state_dict(make_state_dict(cfg, 1234))
net.Network.set_pipelined(True)
n = 16
lr, rf, _ = make_clip(n, 270, 480, seed=0)
lr, rf = lr.to(dev), rf.to(dev)
wins = [window_indices(f, n, 5) for f in range(n)]
wl = [lr[torch.tensor(w, device=dev)][None].contiguous() for w in wins]
wr = [rf[torch.tensor(w, device=dev)][None].contiguous() for w in wins]
for f in range(6):
    net(wl[f], wr[f], f == 0, frame_ids=wins[f])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for f in range(6, 12):
        net(wl[f], wr[f], False, frame_ids=wins[f])
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=6):
    nm = e.key
    if any(k in nm for k in ('copy', 'Memcpy', 'clone', 'fill', 'zero', 'Memset', 'cat', 'stack', 'contiguous', 'to')) and nm.startswith('aten::'):
        rows.append((e.count, nm, [s for s in e.stack if 'refvsr_amd' in s or 'bench' in s][:3]))
rows.sort(reverse=True)
for c, nm, st in rows[:40]:
    print('%5d  %-22s %s' % (c, nm, ' <- '.join(x.strip()[-70:] for x in st)))
