#!/usr/bin/env python3
"""Host-side cost of one steady-state call (pipelined mode): wall per call without device sync + cProfile top."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from refvsr_amd import SRNet, get_config, make_state_dict
from refvsr_amd.synth import make_clip, window_indices
dev = torch.device('cuda:0')
cfg = get_config('b', 'b', 'config_RefVSR_small_L1'); cfg.frame_num = 5
net = SRNet(cfg).to(dev).eval(); net.load_state_dict(make_state_dict(cfg, 1234))
net.Network.set_pipelined(True)
n = 24
lr, rf, _ = make_clip(n, 270, 480, seed=0)
lr, rf = lr.to(dev), rf.to(dev)
wins = [window_indices(f, n, 5) for f in range(n)]
wl = [lr[torch.tensor(w, device=dev)][None].contiguous() for w in wins]
wr = [rf[torch.tensor(w, device=dev)][None].contiguous() for w in wins]
torch.cuda.synchronize()
for f in range(4):
    net(wl[f], wr[f], f == 0, frame_ids=wins[f])
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for f in range(4, n):
    net(wl[f], wr[f], False, frame_ids=wins[f])
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host enqueue per call: %.3f ms (under cProfile); device-complete per call: %.3f ms' % ((t1 - t0) / (n - 4) * 1e3, (t2 - t0) / (n - 4) * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18); print(s.getvalue()[:3500])
