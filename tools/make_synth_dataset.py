#!/usr/bin/env python3
"""Write synthetic clips (refvsr_amd.synth) in the RealMCVSR folder layout the reference reads
(configs/config.py:120-152):  <root>/RealMCVSR/<set>/{LRx4,HR}/{UW,W,T}/<clip>/<frame>.png"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refvsr_amd.evalrun import write_frame  # noqa: E402
from refvsr_amd.synth import make_clip  # noqa: E402


def make(root, clips=2, frames=4, h=32, w=48, test_set='test', hd=False):
    base = os.path.join(root, 'RealMCVSR', test_set)
    lr_dir = 'HR' if hd else 'LRx4'
    for c in range(clips):
        lr, rf, gt = make_clip(frames, h, w, seed=100 + c)
        name = '%04d' % (c + 1)
        for f in range(frames):
            fn = '%04d.png' % f
            write_frame(os.path.join(base, lr_dir, 'UW', name, fn), lr[f])
            write_frame(os.path.join(base, lr_dir, 'W', name, fn), rf[f])
            write_frame(os.path.join(base, lr_dir, 'T', name, fn), rf[f])
            if not hd:
                write_frame(os.path.join(base, 'HR', 'UW', name, fn), gt[f])
    return base


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--root', required=True)
    ap.add_argument('--clips', type=int, default=2)
    ap.add_argument('--frames', type=int, default=4)
    ap.add_argument('--h', type=int, default=32)
    ap.add_argument('--w', type=int, default=48)
    a = ap.parse_args()
    print(make(a.root, a.clips, a.frames, a.h, a.w))
