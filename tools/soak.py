#!/usr/bin/env python3
"""Long-stream soak of the three call modes on one GPU: a synthetic clip of N frames (default 600: 66 reset_branch roll-overs of
config_RefVSR_small_L1) through (a) frame groups of four, (b) one forward(frame_ids=) per frame, pipelined, (c) the reference call
surface, sequential -- every output frame of (a) and (b) must equal (c) bit for bit (compared on the device), and the allocator's
high-water mark must stop growing after the first hundred frames (no leak across roll-overs / cache evictions).  Prints one line per
100 frames and a verdict; exit code 1 on any difference."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from refvsr_amd import SRNet, get_config, make_state_dict  # noqa: E402
from refvsr_amd.synth import window_indices  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=600)
    ap.add_argument('--size', default='270x480')
    ap.add_argument('--config', default='config_RefVSR_small_L1')
    ap.add_argument('--chunk', type=int, default=100, help='frames whose reference outputs are held on the device at a time')
    args = ap.parse_args()
    H, W = [int(v) for v in args.size.split('x')]
    dev = torch.device('cuda:0')
    cfg = get_config('soak', 'soak', args.config)
    T = cfg.frame_num = 5
    sd = make_state_dict(cfg, 1234)
    nets = []
    for _ in range(3):
        n_ = SRNet(cfg).to(dev).eval()
        n_.load_state_dict(sd)
        nets.append(n_)
    ref_net, per_net, grp_net = nets
    per_net.Network.set_pipelined(True)
    grp_net.Network.set_pipelined(True)
    nfr = args.frames
    # frames made on the GPU (the host-side generator of synth.make_clip renders 1080p ground truth per frame: minutes for 600 frames):
    # a smooth random texture scrolling by (1, 2) LR pixels per frame, quantised to 8 bit like decoded frames; the reference camera sees
    # the same scene through another blur
    import torch.nn.functional as F
    g = torch.Generator(device='cpu').manual_seed(0)
    th, tw = H + nfr + 16, W + 2 * nfr + 16
    tex = F.interpolate(torch.rand(1, 3, th // 6 + 2, tw // 6 + 2, generator=g), size=(th, tw), mode='bicubic', align_corners=False)[0].clamp_(0, 1)
    tex = (tex + 0.08 * torch.rand(3, th, tw, generator=g)).clamp_(0, 1).to(dev)
    tex_ref = F.avg_pool2d(tex[None], 3, 1, 1)[0]
    q8 = lambda x: torch.round(x * 255.0) / 255.0

    def frame(f):
        return q8(tex[:, f:f + H, 2 * f:2 * f + W]).contiguous(), q8(tex_ref[:, f:f + H, 2 * f:2 * f + W]).contiguous()

    def window(f):
        idx = window_indices(f, nfr, T)
        lrs = torch.stack([frame(i)[0] for i in idx], 0)[None].contiguous()
        rfs = torch.stack([frame(i)[1] for i in idx], 0)[None].contiguous()
        return lrs, rfs, idx
    bad = 0
    peaks = []
    t0 = time.perf_counter()
    for c0 in range(0, nfr, args.chunk):
        c1 = min(nfr, c0 + args.chunk)
        wins = [window(f) for f in range(c0, c1)]
        torch.cuda.synchronize()
        want = [ref_net(lrs, rfs, f == 0)['result'].clone() for f, (lrs, rfs, _) in zip(range(c0, c1), wins)]
        got_p = [per_net(lrs, rfs, f == 0, frame_ids=idx, input_ready='materialised')['result'] for f, (lrs, rfs, idx) in zip(range(c0, c1), wins)]
        groups = []
        f = c0
        while f < c1:
            n = min(4, c1 - f)
            groups.append((f, n, torch.cat([wins[f - c0 + b][0] for b in range(n)], 0), torch.cat([wins[f - c0 + b][1] for b in range(n)], 0)))
            f += n
        torch.cuda.synchronize()                                        # 'materialised' is a promise about the inputs
        got_g = []
        for f, n, lrs, rfs in groups:                                   # back to back: groups overlap on the internal streams
            got_g += list(grp_net.forward_group(lrs, rfs, [wins[f - c0 + b][2] for b in range(n)], is_first_frame=(f == 0),
                                                input_ready='materialised')['result'])
        torch.cuda.synchronize()
        for i in range(c1 - c0):
            if not torch.equal(got_p[i], want[i]):
                bad += 1
                print('frame %d: one-frame-per-call differs from the sequential reference surface' % (c0 + i), flush=True)
            if not torch.equal(got_g[i], want[i]):
                bad += 1
                print('frame %d: frame group differs from the sequential reference surface' % (c0 + i), flush=True)
        del want, got_p, got_g, wins, groups
        peaks.append(torch.cuda.max_memory_allocated(dev) / 2.0 ** 30)
        print('soak frames %4d..%4d  equal so far: %s  peak allocated %.3f GiB  reserved %.3f GiB  %.1f s' %
              (c0, c1 - 1, bad == 0, peaks[-1], torch.cuda.memory_reserved(dev) / 2.0 ** 30, time.perf_counter() - t0), flush=True)
    grew = len(peaks) > 2 and peaks[-1] > peaks[1] * 1.02
    print('SOAK_RESULT frames=%d config=%s size=%s differences=%d peak_GiB_after_200=%.3f peak_GiB_final=%.3f growing=%s' %
          (nfr, args.config, args.size, bad, peaks[min(1, len(peaks) - 1)], peaks[-1], grew), flush=True)
    sys.exit(1 if (bad or grew) else 0)


if __name__ == '__main__':
    main()
