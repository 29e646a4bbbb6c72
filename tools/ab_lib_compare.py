#!/usr/bin/env python3
"""A/B of two builds of librefvsr_hip.so on the GPU: the matching's small kernels (refvsr_match_patches, refvsr_match_refine) of the
in-tree build against a previous build (default tools/gpu_runs/ab/librefvsr_hip_prev.so), bit for bit on the same inputs, with device
microseconds per launch of both.  Used for kernel rewrites that must not change a single bit of the index map's inputs."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from refvsr_amd import hip, ops  # noqa: E402

dev = torch.device('cuda:0')


def load(path):
    h = C.CDLL(path)
    for name in ('refvsr_match_patches', 'refvsr_match_refine'):
        fn = getattr(h, name)
        fn.argtypes = hip.SIGNATURES[name]
        fn.restype = C.c_int
    return h


def us(fn, iters=20):
    import gc
    gc.collect()
    gc.disable()                         # a gen-2 collection inside the timed loop stalls the host for tens of ms: the GPU
    try:                                 # idles and the events report milliseconds per launch (seen twice in round 5)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    finally:
        gc.enable()


def main():
    prev = load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gpu_runs', 'ab', 'librefvsr_hip_prev.so'))
    cur = hip.lib()
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(3)
    ok = True
    for (h, w) in [(2, 2), (5, 3), (14, 18), (33, 127), (9, 128), (7, 129), (40, 257), (135, 240), (270, 480)]:
        base = F.interpolate(torch.randn(1, 16, h // 4 + 2, w // 4 + 2, generator=g), size=(h, w), mode='bilinear')[0]
        f = (base + 0.2 * torch.randn(16, h, w, generator=g)).to(dev).contiguous()
        n = h * w
        res = []
        for lib in (prev, cur):
            rows = torch.zeros((n + 7, hip.MATCH_KP), dtype=torch.float16, device=dev)
            lo = torch.zeros_like(rows)
            inv = torch.zeros(n, dtype=torch.float32, device=dev)
            assert lib.refvsr_match_patches(P(f), h, w, P(rows), P(inv), P(lo), st()) == 0
            rows2 = torch.zeros_like(rows)
            assert lib.refvsr_match_patches(P(f), h, w, P(rows2), P(inv), None, st()) == 0          # without the lo half
            assert torch.equal(rows, rows2)
            res.append((rows, lo, inv))
        same = all(torch.equal(a.view(torch.int16) if a.dtype == torch.float16 else a.view(torch.int32), b.view(torch.int16) if b.dtype == torch.float16 else b.view(torch.int32))
                   for a, b in zip(res[0], res[1]))
        ok &= same
        line = 'match_patches %4dx%-4d bit-identical=%s' % (h, w, same)
        if n >= 135 * 240:
            rows, lo, inv = res[0]
            t0 = us(lambda: prev.refvsr_match_patches(P(f), h, w, P(rows), P(inv), P(lo), st()))
            t1 = us(lambda: cur.refvsr_match_patches(P(f), h, w, P(rows), P(inv), P(lo), st()))
            line += '   prev %.1f us  new %.1f us' % (t0, t1)
        print(line, flush=True)
    # match_refine on real candidate lists (the fused GEMM's top-2) and on adversarial ones (ties, equal candidates, out-of-range ids)
    for (h, w) in [(20, 28), (34, 50), (64, 96), (270, 480)]:
        base = F.interpolate(torch.randn(1, 16, h // 4 + 2, w // 4 + 2, generator=g), size=(h, w), mode='bilinear')[0]
        lr_f = (base + 0.2 * torch.randn(16, h, w, generator=g)).to(dev).contiguous()
        ref_f = (F.avg_pool2d(base[None], 2)[0] + 0.2 * torch.randn(16, h // 2, w // 2, generator=g)).to(dev).contiguous()
        hr, wr = h // 2, w // 2
        lr_rows, inv_lr = ops.match_patches(lr_f, hip.MATCH_COLBLOCK)
        ref_rows, inv_ref = ops.match_patches(ref_f, hip.MATCH_ROWCHUNK)
        cand, cval = ops.match_top2(ref_rows, hr * wr, lr_rows, h * w, 1)
        cands = [('top-2', cand, cval)]
        c2 = cand.clone()
        c2[::3, 1] = c2[::3, 0]                                   # equal candidates
        c2[1::7, 0] = -5                                           # clamped ids
        c2[2::7, 1] = hr * wr + 9
        cands.append(('adversarial', c2, cval))
        for tag, cd, cv in cands:
            out = []
            for lib in (prev, cur):
                conf = torch.zeros(h * w, dtype=torch.float32, device=dev)
                idx = torch.zeros(h * w, dtype=torch.int32, device=dev)
                fl = torch.zeros(h * w + 1, dtype=torch.int32, device=dev)
                assert lib.refvsr_match_refine(P(lr_f), h, w, P(ref_f), hr, wr, P(inv_lr), P(inv_ref), P(cd), P(cv), 2, 2.5e-4, P(fl), P(conf), P(idx), st()) == 0
                torch.cuda.synchronize()
                nfl = int(fl[0])
                out.append((conf, idx, nfl, torch.sort(fl[1:1 + nfl]).values))
            same = (torch.equal(out[0][0].view(torch.int32), out[1][0].view(torch.int32)) and torch.equal(out[0][1], out[1][1]) and
                    out[0][2] == out[1][2] and torch.equal(out[0][3], out[1][3]))
            ok &= same
            line = 'match_refine %4dx%-4d %-11s bit-identical=%s flagged=%d' % (h, w, tag, same, out[0][2])
            if h * w >= 270 * 480 and tag == 'top-2':
                conf, idx = out[0][0], out[0][1]
                t0 = us(lambda: prev.refvsr_match_refine(P(lr_f), h, w, P(ref_f), hr, wr, P(inv_lr), P(inv_ref), P(cd), P(cv), 2, 2.5e-4, None, P(conf), P(idx), st()))
                t1 = us(lambda: cur.refvsr_match_refine(P(lr_f), h, w, P(ref_f), hr, wr, P(inv_lr), P(inv_ref), P(cd), P(cv), 2, 2.5e-4, None, P(conf), P(idx), st()))
                line += '   prev %.1f us  new %.1f us' % (t0, t1)
            print(line, flush=True)
    print('AB_RESULT', 'all bit-identical' if ok else 'DIFFERENCES', flush=True)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
