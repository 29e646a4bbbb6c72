#!/usr/bin/env python3
"""Headline benchmark: 4x SR output frames/s (BASELINE.json `metric`), steady-state sliding-window inference through
the drop-in SRNet surface on the HIP path.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          # re-launches itself under torch.distributed.run, or (the driver's form):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one output frame.  Inputs (the synthetic LR / Ref clip, every sliding window) are resident in HBM before the
timed region.  Workload (default, N = 1): BASELINE configs[1] -- RefVSR_small_L1, 270x480 -> 1080x1920, frame_num = 5;
`--config config_RefVSR_MFID` is configs[2], `--config config_RefVSR_MFID_8K --size 1080x1920` configs[4] on one GPU.
N > 1: `value` is the STRONG-scaling figure -- ONE 64-frame clip of config_RefVSR_small_MFID (configs[3]) sharded by frame index
over the ranks with the forward-state hand-off and the context exchange over RCCL send/recv (refvsr_amd/shard.py:run_wavefront),
barrier + synchronize on both sides, max over ranks, frames verified against a single-rank run; the exchange-free figure (every
rank K steps on its own reset-aligned shard, per-GPU work fixed) sits beside it as `weak_scaling_shards`.

The stdout line is the COMPACT form of the record (< 6 KB: the driver keeps 8 KB); the complete record goes to --full-json.
What the line carries beside the contract fields:
  value              -- the build's fastest supported call mode: FRAME GROUPS (`forward_group`: --group consecutive output frames per
                        call, named by frame ids, on the engine's internal streams, inputs 'materialised') -- documented EXTENSIONS of
                        the reference's call surface, bit-identical to it; the backward branches of a group run as multi-map launches
  one_frame_per_call -- the same K steps with one `forward(frame_ids=)` per frame (round 4's headline mode)
  dropin_surface     -- the same K steps through the UNMODIFIED reference call surface (`net(x, ref, is_first_frame)`,
                        frames recognised by content, no pipelining): what run.py / eval.py get with the 3-line plug-in
  roofline           -- the TIME-dominant kernel (the fused 24-channel ResBlock, a third of the device time, MFMA-bound): useful FLOPs
                        per launch / mean launch duration from HIP events around every run of blocks in one more pass of the same calls
                        with every internal section on ONE stream; in group mode the launches of record are the multi-map launches
                        (`maps_per_launch`); `traffic` = HBM bytes per launch from two rocprofv3 --pmc child passes of THIS run (FETCH_SIZE,
                        WRITE_SIZE; fallback: profiles/pmc_kernels.json, then `traffic_static` is true); both roofs are stated;
                        configurations whose blocks do not run on that kernel (C = 48 / 36) report the matching kernel here
  roofline_match_top2 -- the fused matching GEMM + arg-max (one launch per frame): algorithmic FLOPs per launch / mean duration from HIP
                        events around every launch inside the timed region
  cpu_baseline       -- the CPU oracle (a port of the reference's algorithm; the reference itself cannot travel) timed on
                        this host: ONE full steady-state forward as the reference executes it, nothing sampled
  whole_path, streams_ms_per_frame, other_configs (configs[2], configs[4] on one GPU), kernels (per-kernel rates, stand-alone),
  wavefront_model_predicted_speedup (N = 1) / wavefront (N > 1: the measured sharded-clip leg) -- one figure each; details in the
  full record
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL's buffer exchange fails with `hipIpcGetMemHandle: invalid argument`
# otherwise); the image exports it -- kept here for launchers that build their own environment.  Before the HIP runtime starts.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F16_TFLOPS = 2500.0          # dense MFMA f16/bf16 peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0             # HBM3E spec peak (6.3 TB/s achievable by a float4 copy), MI355X_MICROARCH.md
SURVEY_DEDUP_TFLOP = {'config_RefVSR_small_L1': 2.490, 'config_RefVSR_small_MFID': 2.490}   # SURVEY.md 8(d), 270x480 t=5
BASELINE_CONFIG = {'config_RefVSR_small_L1': 'configs[1]', 'config_RefVSR_MFID': 'configs[2]',
                   'config_RefVSR_small_MFID': 'configs[3]', 'config_RefVSR_MFID_8K': 'configs[4]'}


def usable_cores():
    """Cores this process may really use: affinity mask and cgroup CPU quota (os.cpu_count() reports
    the whole host, which on a quota-limited container oversubscribes OpenMP badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline_child(config, h, w, t):
    """Runs in a child process (so a slow host can be cut off without losing the GPU number)."""
    from refvsr_amd import get_config, make_state_dict
    from refvsr_amd.synth import make_clip
    from oracle import refvsr_oracle as orc
    ncores = min(usable_cores(), 64)
    torch.set_num_threads(ncores)
    cfg = get_config('bench', 'bench', config)
    cfg.frame_num = t
    sd = make_state_dict(cfg, 1234)
    lr, rf, _ = make_clip(t, h, w, seed=0, want_gt=False)
    o = orc.OracleNetwork(cfg, sd, match_chunk=8192)
    C = cfg.mid_channels                      # forward state of a previous call (values do not influence the timing)
    o.forward_feat_prop_prev = torch.zeros(1, C, h, w)
    o.forward_flow_prev = torch.zeros(1, 2, h, w)
    o.forward_feat_prop_UP_prev = torch.zeros(1, C, 2 * h, 2 * w)
    o.forward_conf_map_prop_prev = torch.zeros(1, 1, h, w)
    o.frame_itr_num = 1
    x, r = lr[:t][None], rf[:t][None]
    with torch.no_grad():
        t0 = time.perf_counter()
        o.forward(x, r, False)
        total = time.perf_counter() - t0
    out = {'value': 1.0 / total, 'unit': 'frames/s', 'cores': ncores, 'kind': 'port',
           'sample': ('1 full steady-state forward as the reference executes it (%d SPyNet calls, %d matchings, %d+1 '
                      'propagation steps, upsampler) at %dx%d t=%d fp32, torch CPU, nothing sampled or scaled: %.2f s'
                      % (2 * (t - 1), t // 2 + 1, t - t // 2, h, w, t, total)),
           'seconds_per_frame': total, 'host_cpu_count': os.cpu_count(), 'usable_cores': usable_cores()}
    print('CPU_BASELINE ' + json.dumps(out), flush=True)


def queued_launch_us(fn, iters, blocker):
    """Device time per launch of fn(): `iters` launches are queued behind a long-running kernel (`blocker`), so the GPU
    executes them back to back and the host launch rate does not enter; HIP events on the launch stream."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    blocker()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def kernel_rooflines(cfg, eng, h, w, dev, live_traffic=None):
    """Per-kernel achieved rates on this GPU, in isolation (after the timed region; never part of `value`)."""
    from refvsr_amd import ops
    C = cfg.mid_channels
    g = torch.Generator().manual_seed(3)
    rnd16 = lambda hh, ww, c: ops.pack_nhwc16(torch.randn(c, hh, ww, generator=g).to(dev))
    x_lr, x_2x = rnd16(h, w, C), rnd16(2 * h, 2 * w, C)
    flow = (torch.randn(2, h, w, generator=g) * 2).to(dev)
    flow2 = ops.flow_up2(flow)
    idx = torch.randint(0, (h // 2) * (w // 2), (h * w,), generator=g, dtype=torch.int32).to(dev)
    aff = (torch.rand(3, h, w, generator=g) * 0.4 + 0.8).to(dev)
    lr = torch.rand(3, h, w, generator=g).to(dev)
    big_a, big_b = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
    blocker = lambda: torch.mm(big_a, big_b)          # ~1 ms of GPU time to queue the measured launches behind
    traffic = dict(live_traffic or {})
    pj = os.path.join(ROOT, 'profiles', 'pmc_kernels.json')
    if not traffic and os.path.exists(pj):
        try:
            traffic = json.load(open(pj)).get('traffic_bytes_per_launch', {})
        except Exception:  # noqa: BLE001
            traffic = {}
    out = []

    def add(name, fn, flops, nbytes, bound, tkey=None, iters=40):
        us = queued_launch_us(fn, iters, blocker)
        tf, gbs = flops / us / 1e6, nbytes / us / 1e3
        ent = {'kernel': name, 'bound': bound, 'us_per_launch': round(us, 2), 'algorithmic_flops': flops, 'algorithmic_bytes': nbytes,
               'tflops': round(tf, 1), 'frac_mfma': round(tf / PEAK_F16_TFLOPS, 4), 'gbs': round(gbs, 1),
               'frac_hbm': round(gbs / PEAK_HBM_GBS, 4), 'traffic': traffic.get(tkey or name)}
        ent['achieved'], ent['peak'], ent['unit'] = (gbs, PEAK_HBM_GBS, 'GB/s') if bound == 'hbm' else (tf, PEAK_F16_TFLOPS, 'TFLOP/s')
        ent['frac'] = ent['achieved'] / ent['peak']
        out.append(ent)

    nb = cfg.num_blocks
    pair = lambda name: (eng.cw(name + '.conv1'), eng.cw(name + '.conv2'))
    c1, c2 = pair('backward_resblocks.main.2.%d' % (nb // 2))
    d1, d2 = pair('feat_decoder2.RBs.1')
    rb_flops = lambda px: 2 * 2.0 * 9 * C * C * px
    rb24 = eng.fuse_resblocks and eng.rb24 and C == 24
    rb_bytes = lambda px: 2.0 * px * C * 2 + (43264 if rb24 else 2 * (c1.wpack.numel() * 2))
    if eng.fuse_resblocks:      # through the engine's own dispatch: the kernel the frame uses (resblock24 for C = 24)
        add('resblock_fused LR (backward_resblocks)', lambda: eng._block_chain(x_lr, [(c1, c2)], 0.0), rb_flops(h * w), rb_bytes(h * w), 'mfma',
            'resblock LR')
        add('resblock_fused 2x (feat_decoder2)', lambda: eng._block_chain(x_2x, [(d1, d2)], 0.2), rb_flops(4 * h * w), rb_bytes(4 * h * w), 'mfma',
            'resblock 2x')
    cw = eng.cw('conv_hr')
    x_hr = rnd16(4 * h, 4 * w, C)
    add('conv_mfma HR %d->%d 3x3 (conv_hr)' % (C, C), lambda: ops.conv(cw, x_hr, act=0.1), 2.0 * 9 * C * C * 16 * h * w,
        2.0 * 16 * h * w * C * 2, 'mfma', 'conv HR')
    cwu = eng.cw('upsample2.upsample_conv')
    add('conv 2x %d->%d 3x3 + pixel shuffle (upsample2)' % (C, 4 * C), lambda: ops.conv(cwu, x_2x, act=0.1), 2.0 * 9 * C * 4 * C * 4 * h * w,
        (4 * h * w * C + 16 * h * w * C) * 2.0, 'mfma', 'conv shuffle 2x')
    wb = lambda px: (2.0 * C * 2 + 8) * px
    add('warp_nhwc16 LR', lambda: ops.warp_nhwc16(x_lr, flow), 0.0, wb(h * w), 'hbm', 'warp LR')
    add('warp_nhwc16 2x', lambda: ops.warp_nhwc16(x_2x, flow2), 0.0, wb(4 * h * w), 'hbm', 'warp 2x')
    # algorithmic bytes of the gathers: every input byte once + every output byte once (re-reads are cache hits)
    add('block_gather_nhwc16 2x (aa2)', lambda: ops.block_gather_nhwc16(x_lr, idx, h, w, 2), 0.0, C * 2.0 * (4 * h * w + h * w) + 4.0 * h * w,
        'hbm', 'gather 2x')
    add('aligned_sample 2x (AlignedConv2d sampler)', lambda: ops.aligned_sample(x_2x, aff, 2), 0.0, 2.0 * C * 2 * 4 * h * w + 12.0 * h * w, 'hbm',
        'aligned_sample 2x')
    add('resize bicubic x4 (base)', lambda: ops.bicubic_scale(lr, 4, clamp01=True), 0.0, 3 * 4.0 * (h * w + 16 * h * w), 'hbm', 'bicubic x4')
    # round 4: the launches that replaced several (algorithmic bytes = inputs once + outputs once)
    add('warp_nhwc16_up2 2x (warp by the up-sampled LR flow, no 2x flow map)', lambda: ops.warp_nhwc16_up2(x_2x, flow), 0.0,
        2.0 * C * 2 * 4 * h * w + 8.0 * h * w, 'hbm', 'warp up2 2x')
    try:
        ca, cb = torch.rand(1, h, w, generator=g).to(dev), torch.rand(1, h, w, generator=g).to(dev)
        w0, b0 = eng.W.raw['conf_fusion2.0.0']
        cwa = eng.cw('conf_fusion2.1.0')
        if ops.conf_alpha_ok(cwa):
            fl_ = lambda px: 2.0 * 9 * (2 * 16 + 16 * C) * px
            add('conf_alpha LR (cat + 2->16 conv + 16->%d conv + max, one launch)' % C, lambda: ops.conf_alpha(ca, cb, 1, w0, b0, cwa, want_max=True),
                fl_(h * w), 8.0 * h * w + 2.0 * C * h * w + 4.0 * h * w, 'hbm', 'conf_alpha LR')
            add('conf_alpha 2x (cat + bicubic x2 + 2->16 conv + 16->%d conv, one launch)' % C, lambda: ops.conf_alpha(ca, cb, 2, w0, b0, cwa),
                fl_(4 * h * w), 8.0 * h * w + 2.0 * C * 4 * h * w, 'hbm', 'conf_alpha 2x')
    except Exception:  # noqa: BLE001
        pass
    return out


def other_config_leg(name, h, w, steps, warmup, dev, repeats=3):
    """One more BASELINE config through the same fast call mode (frame ids + pipelined calls, inputs resident): frames/s
    (median of `repeats` passes), whole-path fraction of the MFMA peak and the roofline of its dominant conv kernel (device time
    per launch from HIP events around back-to-back launches).  Never part of `value`.  The 1080p clip of configs[4] is the
    270x480 synthetic clip upsampled x4 (bicubic, re-quantised to 8 bit) on the GPU: generating 4320x7680 ground truth on the host
    would take a minute per frame."""
    import torch.nn.functional as F
    from refvsr_amd import SRNet, get_config, make_state_dict, ops
    from refvsr_amd.flops import tflop_per_frame
    from refvsr_amd.synth import make_clip, window_indices
    cfg = get_config('bench', 'bench', name)
    cfg.frame_num = t = 5
    net = SRNet(cfg).to(dev).eval()
    net.load_state_dict(make_state_dict(cfg, 1234))
    nfr = warmup + steps
    if (h, w) == (270, 480):
        lr, rf, _ = make_clip(nfr, h, w, seed=0, want_gt=False)
        lr, rf = lr.to(dev), rf.to(dev)
    else:
        assert h % 270 == 0 and w % 480 == 0 and h // 270 == w // 480
        lr0, rf0, _ = make_clip(nfr, 270, 480, seed=0, want_gt=False)
        up = lambda x: torch.round(F.interpolate(x.to(dev), scale_factor=h // 270, mode='bicubic', align_corners=False).clamp_(0, 1) * 255.0) / 255.0
        lr, rf = up(lr0), up(rf0)
    wins = [window_indices(f, nfr, t) for f in range(nfr)]
    win_lr = [lr[torch.tensor(wi, device=dev)][None].contiguous() for wi in wins]
    win_rf = [rf[torch.tensor(wi, device=dev)][None].contiguous() for wi in wins]
    del lr, rf
    torch.cuda.synchronize()
    N = net.Network
    eng = N.ensure_engines(1, dev)[0]
    # call modes: one forward() per frame, and -- for mid_channels = 24 models, where the groups pay (neutral at C = 48, whose
    # streams stay on round 4's layout) -- frame groups of four; `value` = the first mode of the list (decided before measuring)
    # (groups first: an engine driven through forward_group lays its streams out as P | F | M before the first one exists)
    modes = ([4] if (eng.group_ok() and cfg.mid_channels == 24 and h * w <= 4 * 270 * 480) else []) + [1]
    all_lr = torch.cat(win_lr, 0) if 4 in modes else None
    all_rf = torch.cat(win_rf, 0) if 4 in modes else None
    by_mode = {}
    for G in modes:
        fps = []
        for rep in range(repeats + 1):                  # first repetition = warm-up of this model's kernels and pools
            N.reset()
            N.set_pipelined(True)

            def run(f0, f1):
                o, f = None, f0
                while f < f1:
                    n = min(G, f1 - f)
                    if n >= 2:
                        o = net.forward_group(all_lr[f:f + n], all_rf[f:f + n], [wins[f + b] for b in range(n)], is_first_frame=(f == 0),
                                              input_ready='materialised')['result'][-1]
                    else:
                        o = net(win_lr[f], win_rf[f], f == 0, frame_ids=wins[f], input_ready='materialised')['result']
                    f += n
                return o
            run(0, warmup)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = run(warmup, nfr)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            assert bool(torch.isfinite(out).all())
            if rep > 0:
                fps.append(steps / el)
        N.set_pipelined(False)
        fps.sort()
        by_mode[G] = fps
    best = modes[0]                                   # fixed a priori: groups where the engine has them and they pay, else per frame
    fps = by_mode[best]
    value = fps[len(fps) // 2]
    del all_lr, all_rf
    alg, _ = tflop_per_frame(cfg, h, w, t, dedup=True)
    C_ = cfg.mid_channels
    res = {'workload': '%s 4x SR, %dx%d -> %dx%d, frame_num=5, steady state, frame ids + pipelined calls (BASELINE %s)'
                       % (name, h, w, 4 * h, 4 * w, BASELINE_CONFIG.get(name, '-')),
           'value': value, 'unit': 'frames/s', 'samples': [round(v, 2) for v in fps], 'steps': steps, 'warmup': warmup, 'ms_per_step': 1e3 / value,
           'frames_per_call': best, 'by_frames_per_call': {str(g_): round(v[len(v) // 2], 2) for g_, v in by_mode.items()},
           'whole_path': {'algorithmic_tflop_per_frame': alg, 'achieved_tflops': alg * value, 'frac_of_f16_mfma_peak': alg * value / PEAK_F16_TFLOPS},
           'peak_memory_gib': round(torch.cuda.max_memory_allocated(dev) / 2.0 ** 30, 2)}
    try:                                                # dominant kernel: the residual block of the propagation branches on the LR map,
        g = torch.Generator().manual_seed(5)            # through the engine's own dispatch (resblock48 up to 540 x 960, else two conv48 launches)
        x = ops.pack_nhwc16(torch.randn(C_, h, w, generator=g).to(dev))
        k_ = cfg.num_blocks // 2
        pair = (eng.cw('backward_resblocks.main.2.%d.conv1' % k_), eng.cw('backward_resblocks.main.2.%d.conv2' % k_))
        big_a, big_b = torch.randn(4096, 4096, device=dev), torch.randn(4096, 4096, device=dev)
        us = queued_launch_us(lambda: eng._block_chain(x, [pair], 0.0), 20, lambda: torch.mm(big_a, big_b))
        fused = bool(getattr(eng, 'rb48', False)) and C_ == 48 and h * w <= getattr(eng, 'rb48_max_pixels', 0)
        flops = 2 * 2.0 * 9 * C_ * C_ * h * w
        tf = flops / us / 1e6
        res['roofline'] = {'kernel': '%s (conv3x3-ReLU-conv3x3 + residual of the propagation branches, %d channels, LR map %dx%d; %d blocks per frame)'
                                     % ('resblock48_kernel, one launch' if fused else 'conv%d_kernel x 2 launches' % C_, C_, h, w, cfg.num_blocks * (t - t // 2 + 1)),
                           'bound': 'mfma', 'achieved': tf, 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': tf / PEAK_F16_TFLOPS,
                           'traffic': None, 'us_per_block': round(us, 2), 'flops_per_block': flops}
    except Exception as e:  # noqa: BLE001
        res['roofline'] = {'error': repr(e)[:200]}
    del net, win_lr, win_rf
    torch.cuda.empty_cache()
    return res


def wavefront_model(per_frame, nfr, reset_branch, t_msg=0.3, head_fraction=0.24, one_rank_clip_ms=None, one_rank_wavefront_ms=None):
    """serial_fraction and predicted strong-scaling speedups of shard.run_wavefront from measured per-frame phase times
    (shard.simulate_wavefront: pre-emptive makespan model of the two-lane schedule; t_msg = one 33 MB message over one xGMI link +
    latency, 0.3 ms assumed at N = 1, the measured ring time at N > 1: what a context costs; the hand-off goes in two messages and
    the chain only waits for the first -- header + LR state, head_fraction of the bytes -- plus 0.03 ms; cold = extra phase-A time
    of a block's first frame, measured or 0.85 x phase A)."""
    from refvsr_amd import shard
    ta, tb1, tb2 = per_frame['phase_a_ms'], per_frame['phase_b1_ms'], per_frame['phase_b2_ms']
    cold = per_frame.get('phase_a_cold_extra_ms')
    cold = 0.85 * ta if cold is None else cold
    tot = ta + tb1 + tb2
    ex = exchange_terms(per_frame, ctx_ms=t_msg)
    th = head_fraction * t_msg + 0.03                     # chain-critical part of a hand-off (shard.EngineExecutor.split_handoff)
    out = {'serial_fraction': tb1 / tot if tot > 0 else None, 'message_ms': t_msg, 'handoff_ms_on_the_chain': th, 'cold_block_start_ms': cold,
           'cold_block_start_ms_at_a_restart': 2.0 * cold,
           'context_exchange': {'context_prepare_ms': ex['t_prep'], 'cold_window_extra_with_contexts_ms': ex['t_cold_x'], 'message_ms_assumed': ex['t_ctx'],
                                'what': 'shard.run_wavefront(exchange_contexts=True): every per-frame context (matching, reference encoders, aligned '
                                        'attention) prepared ONCE, by the owner of its frame, and sent to the ranks whose windows need it'},
           'predicted_speedup': {}}
    G, ta1 = int(per_frame.get('phase_a_group', 1) or 1), per_frame.get('phase_a_single_ms')
    gk = dict(group=G, t_a_single=ta1)
    out['phase_a_group'] = G
    # Denominators (the speed-ups used to be relative to a rank that walks the clip phase by phase, 153.6 frames/s
    # where the N = 1 headline was 231.6).  `speedup` = against one rank's phase sum at the SAME per-frame times (the model's own
    # unit); `speedup_vs_one_rank_fast_path` = against the wall time ONE GPU needs for the same clip through its fastest call mode
    # (forward_group on three streams; one_rank_clip_ms, measured here) -- the strong-scaling figure a user sees.  The model's ranks
    # execute one task at a time (no overlap between the lanes), so the second figure is a lower bound of what the model stands for.
    # With one_rank_wavefront_ms -- the SAME executor measured with one rank (bench.one_rank_wavefront_seconds) -- the figure is
    # calibrated instead: speedup (the model's, relative to one rank of the same executor) x fast path / one-rank executor time, i.e.
    # the ranks are assumed to overlap their lanes at N ranks as the one rank measurably does.
    base_ms = one_rank_wavefront_ms if one_rank_wavefront_ms else nfr * tot
    out['one_rank'] = {'phase_sum_ms_per_frame': tot, 'phase_sum_clip_ms': nfr * tot, 'fast_path_clip_ms': one_rank_clip_ms,
                       'fast_path_ms_per_frame': (one_rank_clip_ms / nfr) if one_rank_clip_ms else None,
                       'executor_clip_ms': one_rank_wavefront_ms,
                       'lane_overlap_factor (executor / phase sum)': (one_rank_wavefront_ms / (nfr * tot)) if one_rank_wavefront_ms else None,
                       'vs_fast_path_is': 'speedup x fast_path_clip_ms / %s' % ('executor_clip_ms (measured)' if one_rank_wavefront_ms else 'phase_sum_clip_ms')}
    vs_fast = lambda s_: None if not one_rank_clip_ms else round(s_ * one_rank_clip_ms / base_ms, 3)
    sp = lambda n, parts, rb, il=True: round(shard.predicted_speedup(nfr, n, parts, rb, ta, tb1, tb2, th, cold, il, **gk)[0], 3)
    for n in (2, 4, 8):
        bal = shard.partition(nfr, n)
        grow = shard.partition_chain(nfr, n, tb1 / ta if ta > 0 else 0.165)
        ent = {}
        if reset_branch:
            blk, s_, nm = shard.choose_partition(nfr, n, reset_branch, ta, tb1, tb2, th, cold, **gk)
            ent['with_restarts (reset_branch=%d)' % reset_branch] = {'chosen': nm, 'speedup': round(s_, 3), 'speedup_vs_one_rank_fast_path': vs_fast(s_),
                                                                    'hybrid_reset_aligned': sp(n, shard.partition_hybrid(nfr, n, reset_branch), reset_branch),
                                                                    'balanced_handoff_at_every_boundary': sp(n, bal, reset_branch)}
        if reset_branch:
            blk, s_, nm = shard.choose_partition(nfr, n, reset_branch, ta, tb1, tb2, th, cold, exchange=ex, **gk)
            ent['with_restarts (reset_branch=%d)' % reset_branch]['with_context_exchange'] = {'chosen': nm, 'speedup': round(s_, 3),
                                                                                             'speedup_vs_one_rank_fast_path': vs_fast(s_)}
        blk, s_, nm = shard.choose_partition(nfr, n, None, ta, tb1, tb2, th, cold, exchange=ex, **gk)
        ex_free = {'chosen': nm, 'speedup': round(s_, 3), 'speedup_vs_one_rank_fast_path': vs_fast(s_), 'block_sizes': [b_ - a_ for a_, b_, _ in blk]}
        blk, s_, nm = shard.choose_partition(nfr, n, None, ta, tb1, tb2, th, cold, **gk)
        ent['no_restarts (reset_branch=None, configs[4] regime)'] = {
            'with_context_exchange': ex_free,
            'chosen': nm, 'speedup': round(s_, 3), 'balanced': sp(n, bal, None), 'growing_shards': sp(n, grow, None),
            'block_cyclic_3': sp(n, shard.partition_cyclic(nfr, n, 3), None) if 3 * n < nfr else None,
            'growing_shards_round3_order (B1 after ALL local phase A)': sp(n, grow, None, False)}
        out['predicted_speedup'][str(n)] = ent
    return out


PHASE_GROUP = 4          # windows per phase-A group of the sharded executor (Engine.phase_a_group: REFVSR_MAX_MAPS)


def measure_phases(net, cfg, dev, h, w, t=5, group=PHASE_GROUP):
    """Per-frame DEVICE time of the three phases of shard.run_wavefront over one restart unit (or 9 frames) on this GPU: phase A of
    all its frames -- in groups of `group` windows (Engine.phase_a_group: the backward branches as multi-map launches; what the
    sharded executor runs since round 6) and, for the model's partial groups, window by window --, then the forward-branch chain
    (B1), then the upsamplers (B2), everything queued on ONE stream with HIP events between the phases: no host synchronisation
    inside the timed region (round 5 synchronised after every phase of every frame: the launch latency of each phase's first
    kernels was part of its time), no overlap between the phases (the model adds that: simulate_wavefront's lanes).  First
    repetition = warm-up.  Also the extra phase-A time of a COLD window in the middle of a clip (a block start of the wavefront
    partitions)."""
    from refvsr_amd.synth import make_clip, window_indices
    R = cfg.reset_branch or 9
    nfr = R + 4
    lr, rf, _ = make_clip(nfr, h, w, seed=0, want_gt=False)
    lr, rf = lr.to(dev), rf.to(dev)
    N = net.Network
    win = lambda f: (lr[torch.tensor(window_indices(f, nfr, t), device=dev)][None].contiguous(),
                     rf[torch.tensor(window_indices(f, nfr, t), device=dev)][None].contiguous(), window_indices(f, nfr, t))
    wins = [win(f) for f in range(R)]

    def unit(G):
        """One restart unit: A (groups of G) | B1 chain | B2s on the current stream; returns (ms A, ms B1, ms B2)."""
        N.reset()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        ev[0].record()
        hs = []
        for f0 in range(0, R, G):
            fs = list(range(f0, min(f0 + G, R)))
            if G > 1:
                hs += N.phase_a_group([wins[f][0][0] for f in fs], [wins[f][1][0] for f in fs], [wins[f][2] for f in fs], [f == 0 for f in fs])
            else:
                hs.append(N.phase_a(wins[f0][0], wins[f0][1], frame_ids=wins[f0][2], first_hint=(f0 == 0)))
        ev[1].record()
        for f in range(R):
            N.phase_b1(hs[f], f == 0)
        ev[2].record()
        outs = [N.phase_b2(hs[f]) for f in range(R)]
        ev[3].record()
        torch.cuda.synchronize()
        del outs
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
    eng0 = N.ensure_engines(1, dev)[0]
    G = max(1, int(group)) if eng0.group_ok() else 1
    def best(G_):
        """One warm-up unit, then the per-phase MINIMUM over three units: the events bracket device time, and a unit whose host issue
        stalls once (an allocator refill behind N.reset(), a collector pass) carries the gap in the phase it fell into -- seen as
        7.8-10 ms for phase A in one run where the other units of the same process gave 4.5."""
        unit(G_)
        runs = [unit(G_) for _ in range(3)]
        return [min(r_[i] for r_ in runs) for i in range(3)]
    acc = best(G)
    per_frame = {'phase_a_ms': acc[0] / R, 'phase_b1_ms': acc[1] / R, 'phase_b2_ms': acc[2] / R, 'phase_a_group': G,
                 'how': 'device time (HIP events) of A | B1 | B2 of one restart unit of %d frames queued on one stream, phase A in groups of %d windows; '
                        'per-phase minimum over three units after a warm-up unit' % (R, G)}
    if G > 1:
        acc1 = best(1)
        per_frame['phase_a_single_ms'] = acc1[0] / R
    else:
        per_frame['phase_a_single_ms'] = per_frame['phase_a_ms']
    cold = []
    for rep in range(3):                                   # phase A of a mid-clip frame on an empty window cache
        N.reset()
        x, r, ids = win(R // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N.phase_a(x, r, frame_ids=ids, first_hint=False)
        torch.cuda.synchronize()
        cold.append(1e3 * (time.perf_counter() - t0))
    N.reset()
    # (the first-frame call of the restart unit is in the phase-A mean: compare the cold window with the steady frames only)
    per_frame['phase_a_cold_extra_ms'] = max(0.0, min(cold[1:]) - per_frame['phase_a_single_ms'])
    # the context exchange of shard.run_wavefront: one per-frame context prepared on its own (Engine.prepare_context), and phase A of
    # the same cold window when its three contexts are there already (prepared / received ahead): what is left of the cold start
    eng = N.ensure_engines(1, dev)[0]
    prep, coldx = [], []
    for rep in range(3):
        N.reset()
        x, r, ids = win(R // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.prepare_context(x[0, t // 2], r[0, t // 2], (0, ids[t // 2]))       # (engine-level id of frame f of batch element 0)
        torch.cuda.synchronize()
        prep.append(1e3 * (time.perf_counter() - t0))
        for j in range(t // 2 + 1, t):
            eng.prepare_context(x[0, j], r[0, j], (0, ids[j]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N.phase_a(x, r, frame_ids=ids, first_hint=False)
        torch.cuda.synchronize()
        coldx.append(1e3 * (time.perf_counter() - t0))
    N.reset()
    per_frame['context_prepare_ms'] = min(prep[1:])
    per_frame['phase_a_cold_with_contexts_ms'] = min(coldx[1:])
    return per_frame


def exchange_terms(per_frame, ctx_ms=0.3):
    """simulate_wavefront's `exchange` argument from measure_phases: t_prep = one context prepared alone (capped by phase A), t_cold_x =
    what a cold window costs beyond a steady window's remainder once its contexts are there, t_ctx = one 32 MB message over one
    xGMI link + latency (assumed, like the hand-off)."""
    ta = per_frame['phase_a_ms']
    ta1 = per_frame.get('phase_a_single_ms') or ta           # (the cold window is measured as ONE window: compare with a lone steady one)
    tp = min(per_frame.get('context_prepare_ms', 0.35 * ta), ta)
    cx = per_frame.get('phase_a_cold_with_contexts_ms')
    return dict(t_prep=tp, t_ctx=ctx_ms, t_cold_x=max(0.0, cx - (ta1 - tp)) if cx is not None else 0.1 * ta)


def one_rank_clip_seconds(net, cfg, dev, h, w, nfr, t=5, group=PHASE_GROUP, reps=2):
    """Wall seconds ONE GPU needs for the whole nfr-frame clip (cold start, every restart) through its fastest call mode -- frame
    groups of `group` windows on the engine's three streams, inputs resident and materialised: the denominator of every strong-scaling
    figure of this script.  Best of `reps` after one warm-up run; the module is left reset and un-pipelined."""
    from refvsr_amd.synth import make_clip, window_indices
    lr, rf, _ = make_clip(nfr, h, w, seed=0, want_gt=False)
    lr, rf = lr.to(dev), rf.to(dev)
    wins = [window_indices(f, nfr, t) for f in range(nfr)]
    all_lr = torch.stack([lr[torch.tensor(w_, device=dev)] for w_ in wins], 0).contiguous()
    all_rf = torch.stack([rf[torch.tensor(w_, device=dev)] for w_ in wins], 0).contiguous()
    N = net.Network
    eng = N.ensure_engines(1, dev)[0]
    G = max(1, int(group)) if eng.group_ok() else 1
    best = None
    try:
        for rep in range(reps + 1):
            N.reset()
            N.set_pipelined(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f = 0
            while f < nfr:
                n = min(G, nfr - f)
                if n >= 2:
                    o = net.forward_group(all_lr[f:f + n], all_rf[f:f + n], [wins[f + b] for b in range(n)], is_first_frame=(f == 0),
                                          input_ready='materialised')['result'][-1]
                else:
                    o = net(all_lr[f:f + 1], all_rf[f:f + 1], f == 0, frame_ids=wins[f], input_ready='materialised')['result']
                f += n
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            assert bool(torch.isfinite(o).all())
            if rep > 0:
                best = el if best is None else min(best, el)
    finally:
        torch.cuda.synchronize()
        N.set_pipelined(False)
        N.reset()
    return best


def one_rank_wavefront_seconds(net, cfg, dev, h, w, nfr, t=5, group=PHASE_GROUP, reps=2):
    """Wall seconds of shard.run_wavefront over the whole clip with ONE rank (no process group needed): the sharded executor itself
    -- phase-A groups on P | M, the B1 chain on its own lane, upsamplers behind the chains -- with no partner to wait for.  Against one_rank_clip_seconds it says what the executor costs over the single-GPU fast path; against
    the phase sum it gives the overlap of the lanes, which the makespan model (one task at a time per rank) does not have."""
    from refvsr_amd import shard
    from refvsr_amd.synth import make_clip, window_indices
    # (run_wavefront needs no process group for one rank -- and creating a gloo group here would print its banner on STDOUT, ahead of
    #  the one JSON line the driver parses)
    lr, rf, _ = make_clip(nfr, h, w, seed=0, want_gt=False)
    lr, rf = lr.to(dev), rf.to(dev)
    win = {f: (lr[torch.tensor(window_indices(f, nfr, t), device=dev)].contiguous(), rf[torch.tensor(window_indices(f, nfr, t), device=dev)].contiguous())
           for f in range(nfr)}
    ex = shard.EngineExecutor(net, dev, h, w, nfr, t)
    G = max(1, int(group)) if ex.eng.group_ok() else 1
    best = None
    for rep in range(reps + 1):
        net.Network.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = shard.run_wavefront(ex, lambda f: win[f], nfr, t, cfg.reset_branch, cfg.mid_channels, torch.device('cpu'), parts=[(0, nfr)], group=G)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert len(res) == nfr and bool(torch.isfinite(res[nfr - 1]).all())
        del res
        if rep > 0:
            best = el if best is None else min(best, el)
    net.Network.set_pipelined(False)
    net.Network.reset()
    return best


def wavefront_model_single_gpu(args, dev, h, w):
    """N = 1: the phases of BASELINE configs[3] (config_RefVSR_small_MFID, reset_branch 9) timed on this GPU over one restart
    unit, and the speedups the makespan model predicts for 2 / 4 / 8 ranks -- no multi-GPU box needed for the prediction, the
    driver's SCALE run measures the real thing."""
    from refvsr_amd import SRNet, get_config, make_state_dict
    name = 'config_RefVSR_small_MFID'
    cfg = get_config('bench', 'bench', name)
    cfg.frame_num = 5
    net = SRNet(cfg).to(dev).eval()
    net.load_state_dict(make_state_dict(cfg, 1234))
    per_frame = measure_phases(net, cfg, dev, h, w)
    clip_s = one_rank_clip_seconds(net, cfg, dev, h, w, 64)
    out = {'workload': '%s %dx%d, frame_num=5, reset_branch=%d: device time of the phases of one restart unit (9 frames; phase A in groups of %d '
                       'windows) and of a cold mid-clip window on one GPU; prediction for a 64-frame clip (BASELINE configs[3]) under the two-lane '
                       'schedule of shard.run_wavefront, also against the wall time this GPU needs for the same clip through forward_group'
                       % (name, h, w, cfg.reset_branch, per_frame.get('phase_a_group', 1)),
           'phase_ms_per_frame_measured': per_frame,
           'one_rank_same_clip': {'seconds': clip_s, 'frames_per_s': 64.0 / clip_s,
                                  'what': 'the 64-frame clip of configs[3] on ONE GPU, frame groups of 4 on three streams, cold start and the seven restarts included'}}
    try:
        wf1 = one_rank_wavefront_seconds(net, cfg, dev, h, w, 64)
        out['one_rank_wavefront'] = {'seconds': wf1, 'frames_per_s': 64.0 / wf1, 'over_the_fast_path': wf1 / clip_s,
                                     'what': 'shard.run_wavefront over the same clip with ONE rank: the sharded executor (phase-A groups on P | M, '
                                             'B1 lane, upsamplers behind the chains) with nobody to wait for'}
    except Exception as e:  # noqa: BLE001
        wf1 = None
        out['one_rank_wavefront'] = {'error': repr(e)[:300]}
    out.update(wavefront_model(per_frame, 64, cfg.reset_branch, one_rank_clip_ms=1e3 * clip_s, one_rank_wavefront_ms=None if wf1 is None else 1e3 * wf1))
    return out


def run_wavefront_leg(args, rank, world, dev, backend, h, w):
    """BASELINE configs[3]: an args.clip-frame clip of config_RefVSR_small_MFID (reset_branch = 9), sharded by frame index
    over the ranks with the forward-state hand-off (RCCL send/recv of one packed fp16 buffer per block boundary that lies inside
    a restart unit).  Every rank measures its phase times first; the mean decides the partition (shard.choose_partition)."""
    from refvsr_amd import SRNet, get_config, make_state_dict, shard
    from refvsr_amd.synth import make_clip, window_indices
    name = 'config_RefVSR_small_MFID'
    cfg = get_config('bench', 'bench', name)
    cfg.frame_num = t = 5
    nfr = args.clip
    net = SRNet(cfg).to(dev).eval()
    net.load_state_dict(make_state_dict(cfg, 1234))
    comm_dev = dev if backend == 'nccl' else torch.device('cpu')
    pf = measure_phases(net, cfg, dev, h, w)
    keys = ('phase_a_ms', 'phase_b1_ms', 'phase_b2_ms', 'phase_a_cold_extra_ms', 'context_prepare_ms', 'phase_a_cold_with_contexts_ms',
            'phase_a_single_ms')
    v = torch.tensor([pf[k] for k in keys], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(v, op=dist.ReduceOp.SUM)
    per_frame = {k: float(x) / world for k, x in zip(keys, v.cpu().tolist())}
    G = per_frame['phase_a_group'] = int(pf.get('phase_a_group', 1))
    gk = dict(group=G, t_a_single=per_frame['phase_a_single_ms'])
    # the strong-scaling denominator: the SAME clip on one GPU through its fastest call mode, measured by rank 0 while the others wait
    # (ranks that share a GPU -- the one-GPU protocol run -- must not measure under each other)
    clip1 = torch.zeros(1, dtype=torch.float64, device=comm_dev)
    if rank == 0:
        clip1[0] = one_rank_clip_seconds(net, cfg, dev, h, w, nfr, t, reps=1)
    dist.barrier()
    dist.all_reduce(clip1, op=dist.ReduceOp.SUM)
    one_rank_s = float(clip1.item())
    ex = shard.EngineExecutor(net, dev, h, w, nfr, t)
    # warm-up of the point-to-point communicators (their first use costs seconds) and the hand-off time itself: the packed state of
    # this model goes round the ring of ranks, timed on every rank -- BEFORE the partition is chosen: the model's message terms
    # (hand-off, context) are this measurement, not an assumption
    net.Network.reset()
    nb = ex.state_nbytes()
    buf = torch.zeros(nb, dtype=torch.uint8, device=comm_dev)
    handoff_ms = []
    for rep in range(3):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        if world > 1:
            if rank % 2 == 0:
                dist.send(buf, (rank + 1) % world)
                dist.recv(buf, (rank - 1) % world)
            else:
                dist.recv(buf, (rank - 1) % world)
                dist.send(buf, (rank + 1) % world)
        torch.cuda.synchronize()
        handoff_ms.append(1e3 * (time.perf_counter() - t0) / 2.0)      # two messages in series per rank
    hv = torch.tensor([min(handoff_ms[1:])], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(hv, op=dist.ReduceOp.MAX)
    t_msg = max(0.05, float(hv.item()))
    nb_head = ex.state_split_nbytes()[0] if ex.split_handoff else nb
    t_chain = t_msg * nb_head / float(nb) + (0.03 if ex.split_handoff else 0.0)      # what the B1 chain waits for per hand-off
    exch = None if args.no_wavefront_exchange else exchange_terms(per_frame, ctx_ms=t_msg)
    blocks, predicted, pname = shard.choose_partition(nfr, world, cfg.reset_branch, per_frame['phase_a_ms'], per_frame['phase_b1_ms'],
                                                      per_frame['phase_b2_ms'], t_chain, per_frame['phase_a_cold_extra_ms'], exchange=exch, **gk)
    if args.wavefront_partition:                           # A/B: force a partition family
        fam = args.wavefront_partition
        parts = {'balanced': shard.partition(nfr, world), 'growing': shard.partition_chain(nfr, world),
                 'hybrid': shard.partition_hybrid(nfr, world, cfg.reset_branch or 9),
                 'cyclic_growing': shard.partition_cyclic_growing(nfr, world, per_frame['phase_a_ms'], per_frame['phase_b1_ms'] + 0.5 * t_chain,
                                                                  per_frame['phase_a_ms'] + per_frame['context_prepare_ms'])}.get(fam)
        if parts is None and fam.startswith('cyclic'):
            parts = shard.partition_cyclic(nfr, world, int(fam[6:] or 3))
        blocks, pname = shard.as_blocks(parts), fam
        predicted = shard.predicted_speedup(nfr, world, blocks, cfg.reset_branch, per_frame['phase_a_ms'], per_frame['phase_b1_ms'],
                                            per_frame['phase_b2_ms'], t_chain, per_frame['phase_a_cold_extra_ms'], True, exch, **gk)[0]
    mine = [(a, b) for a, b, r in blocks if r == rank]
    need = sorted(set(i for a, b in mine for f in range(a, b) for i in window_indices(f, nfr, t)))
    clip = {}
    for i in need:                                         # this rank's frames (+ input halos), resident in HBM
        if i not in clip:
            l1, r1, _ = make_clip(1, h, w, seed=0, start=i, want_gt=False)
            clip[i] = (l1[0].to(dev), r1[0].to(dev))
    win = {f: (torch.stack([clip[i][0] for i in window_indices(f, nfr, t)], 0).contiguous(),
               torch.stack([clip[i][1] for i in window_indices(f, nfr, t)], 0).contiguous()) for a, b in mine for f in range(a, b)}
    if exch is not None and world > 1:
        # the context messages use their own process group and, with one-frame blocks, rank pairs two apart as well: every pair of
        # the plan exchanges one small message before the clock starts (every rank walks the SAME sorted pair list: a sequence of
        # two-party rendezvous in one global order cannot deadlock)
        plan = shard.ContextPlan(nfr, world, blocks, cfg.reset_branch, t)
        grp = shard.context_group()
        pairs = sorted(set((min(plan.owner[i], q), max(plan.owner[i], q)) for i in range(nfr) for q in plan.consumers[i]))
        tiny = torch.zeros(1024, dtype=torch.uint8, device=comm_dev)
        for a_, b_ in pairs:
            if rank == a_:
                dist.send(tiny, b_, group=grp)
                dist.recv(tiny, b_, group=grp)
            elif rank == b_:
                dist.recv(tiny, a_, group=grp)
                dist.send(tiny, a_, group=grp)
        torch.cuda.synchronize()
    # Two passes of the same sharded clip (round 6): the first, untimed for `value`, warms what the phase measurements cannot -- the
    # caching allocator keeps its pools PER STREAM and the executor's four lanes are new streams: the first pass grows their pools
    # with hipMalloc calls (device-synchronising, milliseconds each) inside a ~50 ms run.  Both passes start from a reset module
    # (cold start of the CLIP -- first frame, empty window cache -- is part of the workload in both); the first one's time is kept.
    el_pass = []
    for rep in range(2):
        net.Network.reset()
        res = None
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        tim = {}
        res = shard.run_wavefront(ex, lambda f: win[f], nfr, t, cfg.reset_branch, cfg.mid_channels, comm_dev, parts=blocks, timings=tim,
                                  exchange_contexts=exch is not None, group=G)
        torch.cuda.synchronize()
        dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        el_pass.append(float(el.item()))
    # per-frame checksums of every rank -> rank 0, compared with rank 0's own sequential run of the first frames
    sums = torch.zeros(nfr, 2, dtype=torch.float64, device=comm_dev)
    for f, r in res.items():
        rd = r.double()
        sums[f, 0], sums[f, 1] = rd.sum().to(comm_dev), (rd * rd).sum().to(comm_dev)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    msgs = torch.tensor([float(tim.get('handoff_messages', 0)), float(tim.get('recv_wait', 0.0)), float(tim.get('context_messages', 0)),
                         float(tim.get('context_wait', 0.0))], dtype=torch.float64, device=comm_dev)
    dist.all_reduce(msgs, op=dist.ReduceOp.SUM)
    out = None
    if rank == 0:
        ncheck = min(nfr, args.clip_check)
        net.Network.reset()
        lr0, rf0, _ = make_clip(min(ncheck + t // 2, nfr), h, w, seed=0, want_gt=False)
        lr0, rf0 = lr0.to(dev), rf0.to(dev)
        ok = True
        for f in range(ncheck):
            # windows of the FULL clip (indices beyond the generated prefix only occur at f >= ncheck)
            wi = torch.tensor(window_indices(f, nfr, t), device=dev)
            r = net(lr0[wi][None], rf0[wi][None], f == 0)['result'][0].double()
            ok = ok and float(r.sum()) == float(sums[f, 0]) and float((r * r).sum()) == float(sums[f, 1])
        C = cfg.mid_channels
        seq_s = nfr * (per_frame['phase_a_ms'] + per_frame['phase_b1_ms'] + per_frame['phase_b2_ms']) * 1e-3
        out = {'ranks_seen': dist.get_world_size(), 'backend': backend + (' (= RCCL)' if backend == 'nccl' else ''),
               'gpus_visible': torch.cuda.device_count(),
               'partition': {'name': pname, 'blocks': [list(b_) for b_ in blocks], 'predicted_speedup': round(float(predicted), 3)},
               'phase_ms_per_frame_measured': per_frame,
               'partition_chosen_with': {'message_ms': t_msg, 'handoff_ms_on_the_chain': t_chain, 'split_handoff': bool(ex.split_handoff),
                                         'what': 'context messages priced at the measured ring time of the packed state; a hand-off at its '
                                                 'first message (header + LR state) when it goes in two'},
               'model': wavefront_model(per_frame, nfr, cfg.reset_branch, t_msg, nb_head / float(nb), one_rank_clip_ms=1e3 * one_rank_s),
               'phase_a_group': G,
               'one_rank_same_clip': {'seconds': one_rank_s, 'frames_per_s': nfr / one_rank_s,
                                      'what': 'the same clip on ONE GPU through forward_group (frame groups of 4, three streams), measured by rank 0 '
                                              'before the sharded run while the other ranks wait'},
               'speedup_vs_n1_headline': one_rank_s / float(el.item()),
               'first_pass_seconds': el_pass[0],
               'passes': 'two passes of the same clip from a reset module; value = the second (the first grows the per-stream allocator pools of the lanes)',
               'workload': '%s, %d-frame clip %dx%d -> %dx%d, frame_num=5, reset_branch=%d, sharded by frame index over %d ranks '
                           '(BASELINE configs[3])' % (name, nfr, h, w, 4 * h, 4 * w, cfg.reset_branch, world),
               'value': nfr / float(el.item()), 'unit': 'frames/s', 'seconds': float(el.item()), 'scaling': 'strong',
               'speedup_over_one_rank_phase_sum': seq_s / float(el.item()),
               'schedule': 'two lanes per rank (HIP streams): phase A (flows, matching, encoders, alignment, backward branch) of the local '
                           'frames on lane 1 in groups of %d windows (backward branches as multi-map launches); the B1 chain (forward-branch steps) '
                           'on lane 2, B1(f) as soon as the group of frame f is done and the state has arrived -- issued between the groups '
                           'wherever it needs no blocking receive --, state sent right after a block\'s last B1; B2 (BW/FW fusion + upsampler) '
                           'on lane 1 one group behind its B1' % G,
               'handoff': {'messages': int(msgs[0].item()), 'bytes_per_message': 64 + h * w * (10 * C + 12),
                           'format': ('two messages: [header | fp16 HWC feat | fp32 flow + conf] (%d bytes: the receiver\'s forward-branch step starts '
                                      'on it), then fp16 HWC feat_up as it lies' % nb_head) if ex.split_handoff else
                                     'one packed buffer: fp16 HWC feat + feat_up, fp32 flow + conf',
                           'ms_per_message_measured': float(hv.item()),
                           'how': 'the packed state of this model sent round the ring of ranks (send / recv pairs), host clock around two '
                                  'messages in series after device synchronisation, best of 2 after one warm-up, max over ranks',
                           'host_seconds_blocked_in_recv_all_ranks': float(msgs[1].item())},
               'context_exchange': None if exch is None else {
                   'messages': int(msgs[2].item()), 'bytes_per_message': ex.context_nbytes() if ex._ctx_spec is not None else None,
                   'format': 'one packed buffer: conf fp32 + index map int32 + the two aligned-attention maps fp16 HWC',
                   'host_seconds_blocked_waiting_all_ranks': float(msgs[3].item()), 'model_terms_ms': exch,
                   'what': 'every per-frame context prepared once, by the owner of its frame, sent point to point (own process group) to '
                           'the ranks whose windows need it; --no-wavefront-exchange: every rank prepares what its windows need'},
               'frames_checked_against_single_rank_run': ncheck, 'frames_equal': bool(ok)}
    dist.barrier()
    return out


def live_pmc_traffic(timeout_s=90.0):
    """HBM bytes per launch measured IN THIS RUN (round 6): the two counter passes MI355X_MICROARCH.md prescribes --
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate child processes, no tracing beside the counters -- on
    tools/pmc_kernels.py (the launches of the `kernels` object + match_top2 + the multi-map block launches at 270 x 480), condensed by
    tools/pmc_to_json.py (FETCH_SIZE x 2 for 16-byte coalesced reads on gfx950, KiB units).  Returns (pmc_kernels dict, pmc_match_top2
    dict) or raises; ~10 s per pass.  The profiler attaches to CHILD processes: nothing of this process is traced."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        raise RuntimeError('rocprofv3 not found')
    nested = [k for k in os.environ if k.startswith(('ROCPROFILER_', 'ROCPROF_', 'ROCP_')) or (k == 'LD_PRELOAD' and 'rocprof' in os.environ[k])]
    if nested:
        # this process is itself being profiled (e.g. `rocprofv3 --kernel-trace -- python bench.py`): a profiler inside a profiled
        # process tree would inherit the outer tool's environment -- the committed counter files serve instead
        raise RuntimeError('already running under a profiler (%s): live counter passes skipped' % ', '.join(sorted(nested)[:3]))
    tools = os.path.join(ROOT, 'tools')
    work = tempfile.mkdtemp(prefix='refvsr_pmc_', dir='/tmp')
    env = dict(os.environ)
    env['TMPDIR'] = '/tmp'
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            r = subprocess.run([exe, '--pmc', ctr, '--output-format', 'csv', '-d', os.path.join(work, ctr), '-o', 'k', '--', sys.executable,
                                os.path.join(tools, 'pmc_kernels.py')], cwd='/tmp', env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                raise RuntimeError('rocprofv3 --pmc %s failed (rc=%d): %s' % (ctr, r.returncode, (r.stderr or r.stdout)[-200:]))
        r = subprocess.run([sys.executable, os.path.join(tools, 'pmc_to_json.py'), os.path.join(work, 'FETCH_SIZE'), os.path.join(work, 'WRITE_SIZE'), work],
                           cwd='/tmp', env=env, capture_output=True, text=True, timeout=60)
        if r.returncode != 0:
            raise RuntimeError('pmc_to_json failed: %s' % r.stderr[-200:])
        return json.load(open(os.path.join(work, 'pmc_kernels.json'))), json.load(open(os.path.join(work, 'pmc_match_top2.json')))
    finally:
        shutil.rmtree(work, ignore_errors=True)


def pcie_inclusive_pass(net, args, all_lr, all_rf, wins, start, G, dev, passes=3, result_dtype='float32', frames_once=False):
    """The K timed steps with host-resident inputs and outputs (see the call site).  Returns the `pcie_inclusive` object of the line.
    result_dtype: config.result_dtype for these passes ('uint8': the output head stores rint(255 v), REFVSR_RESULT_U8 -- the bytes the
    reference's PNG writer makes of the fp32 frame on the CPU -- and a quarter of the bytes goes device -> host)."""
    # frames_once: a streaming caller's loader -- every frame of the clip crosses PCIe ONCE (3.1 MB per output frame instead of the
    # 15.6 MB of whole windows, whose frames overlap t - 1 to t) into a device-side table and the windows are gathered there
    nfr = args.warmup + args.steps
    eng_ = net.Network.ensure_engines(1, dev)[0]
    prev_dtype, eng_.result_dtype = eng_.result_dtype, result_dtype
    table = {}
    h_lr, h_rf = all_lr.cpu().pin_memory(), all_rf.cpu().pin_memory()               # [nfr, t, 3, h, w] fp32, as the reference's loader makes them
    h_out = [None]                                                                   # [nfr, 3, s h, s w] pinned, sized by the first result
    cp = torch.cuda.Stream(dev)
    G = max(1, G)

    def load(f, f1):
        """Host -> device copies of the windows of the call that starts at frame f, on the copy stream; returns what the call needs."""
        n = min(G, f1 - f)
        with torch.cuda.stream(cp):
            if frames_once:
                for b in range(n):
                    for j, i in enumerate(wins[f + b]):
                        if i not in table:
                            table[i] = (h_lr[f + b, j].to(dev, non_blocking=True), h_rf[f + b, j].to(dev, non_blocking=True))
                lr_d = torch.stack([torch.stack([table[i][0] for i in wins[f + b]], 0) for b in range(n)], 0)
                rf_d = torch.stack([torch.stack([table[i][1] for i in wins[f + b]], 0) for b in range(n)], 0)
                for i in [k for k in table if k < wins[f][0]]:
                    del table[i]
            else:
                lr_d = h_lr[f:f + n].to(dev, non_blocking=True)
                rf_d = h_rf[f:f + n].to(dev, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record(cp)
        return n, lr_d, rf_d, ready

    def run(f0, f1):
        # a double-buffered loader: the copies of call k + 1 are issued BEFORE call k (they never queue behind a wait for a result)
        f = f0
        nxt = load(f, f1)
        while f < f1:
            n, lr_d, rf_d, ready = nxt
            nxt = load(f + n, f1) if f + n < f1 else None
            ids = [[start + i for i in wins[f + b]] for b in range(n)]
            if n >= 2:
                res = net.forward_group(lr_d, rf_d, ids, is_first_frame=(f == 0), input_ready=ready)['result']
            else:
                res = [net(lr_d, rf_d, f == 0, frame_ids=ids[0], input_ready=ready)['result']]
            if h_out[0] is None:
                h_out[0] = torch.empty((nfr,) + tuple(res[0].shape[-3:]), dtype=res[0].dtype).pin_memory()
            for b in range(n):
                h_out[0][f + b].copy_(res[b].reshape(h_out[0].shape[1:]), non_blocking=True)
            f += n

    secs = []
    for rep in range(passes + 1):
        net.Network.reset()
        net.Network.set_pipelined(True)
        table.clear()
        run(0, args.warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.warmup, nfr)
        torch.cuda.synchronize()
        if rep:                                                                      # (pass 0: untimed)
            secs.append(time.perf_counter() - t0)
    eng_.result_dtype = prev_dtype
    assert bool(torch.isfinite(h_out[0][args.warmup:].float()).all())
    secs.sort()
    el = secs[len(secs) // 2]
    win_mb = 2 * h_lr[0].numel() * 4 / 1e6
    return {'value': args.steps / el, 'unit': 'frames/s', 'ms_per_step': 1e3 * el / args.steps, 'samples': [round(args.steps / s, 2) for s in secs],
            'h2d_mb_per_frame': round(win_mb, 2), 'd2h_mb_per_frame': round(h_out[0][0].numel() * h_out[0].element_size() / 1e6, 2), 'frames_per_call': G,
            'result_dtype': result_dtype,
            'note': 'windows and results in pinned host memory: every call copies its whole windows (t LR + t reference frames, fp32) host -> device '
                    'on a copy stream and its results (%s) device -> host; host clock to the last result in host memory' % result_dtype}


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def compact_line(line, limit=5600):
    """The stdout form of the record: the contract fields + roofline, cpu_baseline, the drop-in / one-frame-per-call rates and one
    figure per extra leg, < 6 KB (the driver keeps the last 8 KB of stdout; round 4's 14 KB line lost its tail).  The complete record
    goes to --full-json."""
    pick = lambda d, ks: None if not isinstance(d, dict) else {k: _r(d[k]) for k in ks if k in d and d[k] is not None}
    out = {k: line.get(k) for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                                    'vs_baseline', 'dtype', 'data')}
    cfg_ = dict(line.get('config') or {})
    cfg_.pop('precision', None)
    out['config'] = cfg_
    out['samples'] = line.get('samples')
    out['roofline'] = pick(line.get('roofline'), ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'governing', 'mfma_frac', 'mfma_issued_frac',
                                                  'hbm_frac', 'arithmetic_intensity', 'algorithmic_bytes_per_launch', 'traffic', 'traffic_static',
                                                  'maps_per_launch', 'mean_launch_ms', 'launches_timed', 'flops_per_launch', 'error'))
    out['cpu_baseline'] = pick(line.get('cpu_baseline'), ('value', 'unit', 'cores', 'kind', 'sample', 'seconds_per_frame', 'gpu_over_cpu', 'error'))
    if out['cpu_baseline'] and 'sample' in out['cpu_baseline']:
        out['cpu_baseline']['sample'] = str(out['cpu_baseline']['sample'])[:200]
    out['dropin_surface'] = pick(line.get('dropin_surface'), ('value', 'unit', 'samples'))
    out['one_frame_per_call'] = pick(line.get('one_frame_per_call'), ('value', 'unit', 'samples'))
    out['first_frame_ms'] = _r(line.get('first_frame_ms'), 2)
    out['pcie_inclusive'] = pick(line.get('pcie_inclusive'), ('value', 'unit', 'samples', 'h2d_mb_per_frame', 'd2h_mb_per_frame', 'error'))
    if isinstance((line.get('pcie_inclusive') or {}).get('result_uint8'), dict):
        out['pcie_inclusive']['result_uint8'] = pick(line['pcie_inclusive']['result_uint8'], ('value', 'd2h_mb_per_frame'))
        if isinstance(line['pcie_inclusive'].get('result_uint8_frames_once'), dict):
            out['pcie_inclusive']['result_uint8_frames_once'] = pick(line['pcie_inclusive']['result_uint8_frames_once'], ('value', 'h2d_mb_per_frame'))
    out['roofline_match_top2'] = pick(line.get('roofline_match_top2'), ('achieved', 'frac', 'mean_launch_ms', 'traffic'))
    out['whole_path'] = pick(line.get('whole_path'), ('algorithmic_tflop_per_frame', 'achieved_tflops_per_gpu', 'frac_of_f16_mfma_peak',
                                                      'frac_of_f16_mfma_peak_on_survey_figure'))
    st = (line.get('streams') or {}).get('median_pass')
    out['streams_ms_per_frame'] = pick(st, ('P_ms_per_call', 'F_ms_per_call', 'M_ms_per_call', 'wall_ms_per_call'))
    oc = line.get('other_configs')
    if isinstance(oc, dict):
        out['other_configs'] = {k: (pick(v, ('value', 'ms_per_step', 'peak_memory_gib', 'error')) or {}) for k, v in oc.items()}
        for k, v in oc.items():
            if isinstance(v, dict) and isinstance(v.get('whole_path'), dict):
                out['other_configs'][k]['frac_of_f16_mfma_peak'] = _r(v['whole_path'].get('frac_of_f16_mfma_peak'))
            if isinstance(v, dict) and isinstance(v.get('roofline'), dict):
                out['other_configs'][k]['roofline_frac'] = _r(v['roofline'].get('frac'))
    wf = line.get('wavefront')
    if isinstance(wf, dict):
        out['wavefront'] = pick(wf, ('value', 'unit', 'seconds', 'scaling', 'ranks_seen', 'backend', 'gpus_visible', 'frames_equal',
                                     'frames_checked_against_single_rank_run', 'speedup_over_one_rank_phase_sum', 'speedup_vs_n1_headline',
                                     'phase_a_group', 'error'))
        if isinstance(wf.get('one_rank_same_clip'), dict):
            out['wavefront']['one_rank_same_clip_frames_per_s'] = _r(wf['one_rank_same_clip'].get('frames_per_s'))
        if isinstance(wf.get('partition'), dict):
            out['wavefront']['partition'] = wf['partition'].get('name')
            out['wavefront']['predicted_speedup'] = wf['partition'].get('predicted_speedup')
        if isinstance(wf.get('handoff'), dict):
            out['wavefront']['handoff'] = pick(wf['handoff'], ('messages', 'bytes_per_message', 'ms_per_message_measured'))
        if isinstance(wf.get('context_exchange'), dict):
            out['wavefront']['context_exchange'] = pick(wf['context_exchange'], ('messages', 'bytes_per_message'))
    if isinstance(line.get('weak_scaling_shards'), dict):
        out['weak_scaling_shards'] = pick(line['weak_scaling_shards'], ('value', 'unit', 'ms_per_step', 'scaling', 'samples'))
    wm = line.get('wavefront_model')
    if isinstance(wm, dict) and isinstance(wm.get('predicted_speedup'), dict):
        ps = {}
        for n_, ent in wm['predicted_speedup'].items():
            row = {}
            for k, v in ent.items():
                if isinstance(v, dict):
                    x = v.get('with_context_exchange', v)
                    key = 'restarts' if k.startswith('with_restarts') else 'no_restarts'
                    row[key] = x.get('speedup') if isinstance(x, dict) else None
                    if isinstance(x, dict) and x.get('speedup_vs_one_rank_fast_path') is not None:
                        row[key + '_vs_n1_fast_path'] = x.get('speedup_vs_one_rank_fast_path')
            ps[n_] = row
        out['wavefront_model_predicted_speedup'] = ps
        out['wavefront_model_note'] = ('x = against one rank walking the clip phase by phase at the measured per-frame phase times; *_vs_n1_fast_path = '
                                       'against the wall time ONE GPU needs for the same 64-frame clip through forward_group (%s ms per frame)'
                                       % _r(((wm.get('one_rank') or {}).get('fast_path_ms_per_frame')), 3))
        pm = wm.get('phase_ms_per_frame_measured')
        if isinstance(pm, dict):
            out['wavefront_phase_ms_per_frame'] = pick(pm, ('phase_a_ms', 'phase_a_single_ms', 'phase_b1_ms', 'phase_b2_ms', 'phase_a_group'))
    ks = line.get('kernels')
    if isinstance(ks, list):
        out['kernels'] = [[str(k.get('kernel'))[:28], k.get('us_per_launch'), _r(k.get('frac'), 3)] for k in ks]
    out['full_record'] = line.get('full_record')
    for drop in ('kernels', 'wavefront_model_predicted_speedup', 'streams_ms_per_frame', 'roofline_match_top2', 'other_configs'):
        if len(json.dumps(out)) <= limit:
            break
        out.pop(drop, None)
    return out


def emit(line, args):
    """Complete record -> --full-json (best effort), compact line -> stdout."""
    try:
        os.makedirs(os.path.dirname(os.path.abspath(args.full_json)), exist_ok=True)
        with open(args.full_json, 'w') as f:
            json.dump(line, f)
        line['full_record'] = os.path.relpath(args.full_json, ROOT)
    except OSError:
        line['full_record'] = None
    print(json.dumps(line if args.verbose_line else compact_line(line)), flush=True)


def promote_wavefront(line, wf, args, world):
    """N > 1: `value` is the STRONG-scaling figure north_star asks for -- ONE clip of args.clip frames (BASELINE configs[3]) sharded
    over the ranks by frame index with the forward-state hand-off (shard.run_wavefront) -- when that leg ran and its frames equal the
    single-rank run; the exchange-free reset-aligned K-step figure (per-GPU work fixed: it reads ~N x) moves to `weak_scaling_shards`."""
    line['wavefront'] = wf
    if not (isinstance(wf, dict) and wf.get('value') and wf.get('frames_equal')):
        line['config']['headline'] = 'weak-scaling shards (the sharded-clip leg did not produce a verified figure)'
        return
    line['weak_scaling_shards'] = {k: line.get(k) for k in ('value', 'unit', 'ms_per_step', 'samples', 'scaling')}
    line['weak_scaling_shards']['what'] = ('every rank runs the K steps on its own reset-aligned shard of one long clip: no data-path collective, '
                                           'per-GPU work fixed')
    hh = int(args.size.lower().split('x')[0])
    line['metric'] = '4x SR frames/sec (%dp->%dp, RefVSR_small_MFID, %d-frame clip sharded over %d GPUs)' % (hh, 4 * hh, args.clip, world)
    line['value'], line['ms_per_step'], line['scaling'] = wf['value'], 1e3 * wf['seconds'] / float(args.clip), 'strong'
    line['samples'] = [round(wf['value'], 2)]
    cfg_ = line['config']
    cfg_['workload'] = wf['workload']
    cfg_['parallelism'] = ('frame-shard x%d of ONE clip: phase A (flows, matching, encoders, alignment, backward branch) in parallel, the '
                           'forward-branch chain rank to rank with the state hand-off (send/recv), contexts prepared once and exchanged' % world)
    cfg_['timed_frames'] = args.clip
    cfg_['ranks_seen'], cfg_['backend'], cfg_['gpus_visible'] = wf.get('ranks_seen'), wf.get('backend'), wf.get('gpus_visible')
    cfg_['handoff_ms_measured'] = (wf.get('handoff') or {}).get('ms_per_message_measured')
    cfg_['frames_equal_single_rank_run'] = wf.get('frames_equal')
    cfg_['steps_note'] = '--steps / --warmup apply to weak_scaling_shards; the headline times the whole %d-frame clip once' % args.clip


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--repeats', type=int, default=5, help='timed repetitions of the K steps per call mode, interleaved; value = median')
    ap.add_argument('--warm-seconds', type=float, default=0.6, help='untimed whole passes of both call modes before the first timed pass')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the configs[2] / configs[4] legs (other_configs)')
    ap.add_argument('--config', default='config_RefVSR_small_L1')
    ap.add_argument('--size', default='270x480', help='LR frame size HxW (1080x1920 for the 8K configs)')
    ap.add_argument('--frames', type=int, default=5, help='sliding-window length (frame_num)')
    ap.add_argument('--no-cache', action='store_true', help='execute exactly the work the reference executes')
    ap.add_argument('--no-frame-ids', action='store_true', help='headline through the plain call surface (content compare)')
    ap.add_argument('--no-pipeline', action='store_true', help='do not overlap consecutive calls on internal streams')
    ap.add_argument('--group', type=int, default=4, help='output frames per forward_group call of the headline mode (multi-map launches of '
                                                         'the backward branches, 2..4); 1 = one forward() per frame, the round-4 headline')
    ap.add_argument('--full-json', default=os.path.join(ROOT, 'gpurun_out', 'bench_full.json'),
                    help='where the complete record goes (the stdout line is the compact form of it)')
    ap.add_argument('--verbose-line', action='store_true', help='print the complete record on stdout instead of the compact line')
    ap.add_argument('--no-dropin', action='store_true', help='skip the second timed pass through the unmodified call surface')
    ap.add_argument('--no-kernels', action='store_true', help='skip the per-kernel roofline measurements')
    ap.add_argument('--force-dist', action='store_true', help=argparse.SUPPRESS)     # test aid: process group + sharded-clip leg with ONE rank (RCCL on a 1-GPU box)
    ap.add_argument('--no-live-pmc', action='store_true', help='take `traffic` from profiles/pmc_*.json instead of two rocprofv3 --pmc child passes in this run')
    ap.add_argument('--no-wavefront', action='store_true', help='N > 1: skip the sharded-clip leg with the state hand-off')
    ap.add_argument('--clip', type=int, default=64, help='N > 1: frames of the sharded clip (BASELINE configs[3]: 64)')
    ap.add_argument('--clip-check', type=int, default=12, help='frames of the sharded clip re-run on one rank and compared')
    ap.add_argument('--wavefront-timeout', type=float, default=240.0)
    ap.add_argument('--no-wavefront-exchange', action='store_true', help='N > 1: every rank prepares all contexts its windows need (round-4 call 1 schedule)')
    ap.add_argument('--wavefront-partition', default=None, help='N > 1 A/B: balanced | growing | hybrid | cyclicK | cyclic_growing (default: shard.choose_partition)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gc-freeze', action='store_true', help='A/B: leave the collector\'s old generations unfrozen (what a plain eval loop runs with)')
    ap.add_argument('--match-margin', type=float, default=None, help='A/B knob: margin of the exact-search flagging (0 = top-2 re-rank only)')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline-timeout', type=float, default=300.0)
    args = ap.parse_args()
    H, W_ = [int(v) for v in args.size.lower().split('x')]
    T = args.frames
    if args.cpu_baseline_only:
        cpu_baseline_child(args.config, H, W_, T)
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: re-launch under torch.distributed.run, one rank per GPU (the form the driver uses for N > 1,
        # run.py:209-216 in the reference).  Fewer GPUs than ranks (a 1-GPU test box): the ranks share the GPUs and talk over gloo --
        # RCCL refuses two ranks on one device -- and the line says so (`backend`, `gpus_visible`).
        import socket
        s_ = socket.socket()
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
        s_.close()
        if torch.cuda.device_count() < args.gpus:
            os.environ.setdefault('REFVSR_DIST_BACKEND', 'gloo')
        os.environ.setdefault('OMP_NUM_THREADS', '4')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d under a launcher with WORLD_SIZE=%d: launch with --nproc-per-node %d' % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # REFVSR_DIST_BACKEND=gloo lets the N>1 code path be exercised on a single-GPU box (all ranks share GPU 0);
    # the real multi-GPU run uses nccl (= RCCL), one rank per GPU.
    backend = os.environ.get('REFVSR_DIST_BACKEND') or ('gloo' if world > torch.cuda.device_count() else 'nccl')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    # --force-dist (test aid): the process group and the sharded-clip leg also for ONE rank -- on a 1-GPU box that is the only way to
    # run the nccl (= RCCL) branch of this script at all: device-side all_reduce / barrier, communicator set-up, comm_dev = the GPU
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1 and 'MASTER_PORT' not in os.environ:
            import socket
            s_ = socket.socket()
            s_.bind(('127.0.0.1', 0))
            os.environ['MASTER_PORT'] = str(s_.getsockname()[1])
            s_.close()
        dist.init_process_group(backend, rank=rank, world_size=world)

    from refvsr_amd import SRNet, get_config, make_state_dict
    from refvsr_amd.flops import tflop_per_frame
    from refvsr_amd.synth import make_clip, window_indices
    cfg = get_config('bench', 'bench', args.config)
    cfg.frame_num = T
    cfg.cache_windows = not args.no_cache
    if args.match_margin is not None:
        cfg.match_exact_margin = args.match_margin
    sd = make_state_dict(cfg, 1234)
    net = SRNet(cfg).to(dev).eval()
    net.load_state_dict(sd)

    nfr = args.warmup + args.steps
    if nfr > 600:
        raise SystemExit('bench.py keeps the whole synthetic clip and its sliding windows resident (host + HBM): --warmup + --steps <= 600')
    R = cfg.reset_branch or nfr
    start = rank * int(math.ceil(nfr / float(R))) * R          # reset-aligned shard start (exchange-free)
    lr, rf, _ = make_clip(nfr, H, W_, seed=0, start=start, want_gt=False)
    lr, rf = lr.to(dev), rf.to(dev)                             # inputs resident in HBM
    # the sliding windows are materialised before the timed region (inputs resident in HBM)
    wins = [window_indices(f, nfr, T) for f in range(nfr)]
    all_lr = torch.stack([lr[torch.tensor(w, device=dev)] for w in wins], 0).contiguous()       # [nfr, t, 3, h, w]
    all_rf = torch.stack([rf[torch.tensor(w, device=dev)] for w in wins], 0).contiguous()
    win_lr = [all_lr[f:f + 1] for f in range(nfr)]
    win_rf = [all_rf[f:f + 1] for f in range(nfr)]
    torch.cuda.synchronize()
    eng = net.Network.ensure_engines(1, dev)[0]

    def timed_pass(use_ids, pipelined, collect_events, collect_chain=False, timed=True, group=1):
        """W untimed + K timed steps of a new clip; returns (seconds for the K steps, events).  group > 1 (pipelined mode): the
        steps go through forward_group, `group` consecutive output frames per call (the same frames, the same results)."""
        net.Network.reset()
        net.Network.set_pipelined(bool(use_ids and pipelined))
        # pipelined calls: the windows were materialised (and synchronised) before the first pass -- the caller-side assertion
        # that lets the engine's internal streams read them without waiting for the caller's stream (Engine.set_pipelined)
        ready = 'materialised' if (use_ids and pipelined) else None

        def step(f):
            ids = [start + i for i in wins[f]] if use_ids else None
            if ids is None:
                return net(win_lr[f], win_rf[f], f == 0)['result']                  # the reference's call, verbatim
            return net(win_lr[f], win_rf[f], f == 0, frame_ids=ids, input_ready=ready)['result']
        grouped = group > 1 and use_ids and pipelined

        def run_steps(f0, f1):
            o, f = None, f0
            while f < f1:
                n = min(group, f1 - f) if grouped else 1
                if n >= 2:
                    ids = [[start + i for i in wins[f + b]] for b in range(n)]
                    o = net.forward_group(all_lr[f:f + n], all_rf[f:f + n], ids, is_first_frame=(f == 0), input_ready=ready)['result'][-1]
                else:
                    o = step(f)
                f += n
            return o
        out = run_steps(0, args.warmup)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.kernel_events = [] if collect_events else None                           # HIP events around match_top2 ...
        # ... and around the fused-ResBlock runs -- only in the single-stream pass below (16 event records per frame that the
        # timed pass has no use for)
        eng.chain_events = [] if (collect_events and collect_chain) else None
        eng.stream_events = [] if (use_ids and pipelined) else None                  # 6 events per call: the P / F / M sections
        t0 = time.perf_counter()
        out = run_steps(args.warmup, nfr)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1 and timed:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        assert bool(torch.isfinite(out).all())
        ev = (eng.kernel_events, eng.chain_events, eng.stream_events)
        eng.kernel_events = eng.chain_events = eng.stream_events = None
        net.Network.set_pipelined(False)
        return elapsed, ev

    def stream_summary(sev, elapsed):
        """Per-stream busy time of one pipelined pass from the section events (P: per-frame preparation + flows, F: forward-branch
        step, M: backward branch + upsampler): ms per call between each section's first and last kernel, and the span from the
        first P section to the last M section.  In the slow mode of round 3 every section stretches (three kernels share the
        chip); here the sections of a slow pass and of a fast pass can be compared in the line itself."""
        if not sev:
            return None
        out = {}
        nfrm = sum(int(c.get('n', 1)) for c in sev)
        for k in ('P', 'F', 'M'):
            ms = [c[k + '0'].elapsed_time(c[k + '1']) for c in sev if c.get(k + '0') is not None and c.get(k + '1') is not None]
            if ms:
                out[k + '_ms_per_call'] = round(sum(ms) / max(nfrm, 1), 3)        # per OUTPUT FRAME (a group call carries several)
                out[k + '_ms_max'] = round(max(ms), 3)
        first = next((c for c in sev if c.get('P0') is not None), None)
        if first is not None and sev[-1].get('M1') is not None:
            out['span_ms'] = round(first['P0'].elapsed_time(sev[-1]['M1']), 3)
        out['calls'] = len(sev)
        out['frames'] = nfrm
        out['wall_ms_per_call'] = round(1e3 * elapsed / max(nfrm, 1), 3)
        return out

    def med(v):
        v = sorted(v)
        n = len(v)
        return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])

    use_ids = not args.no_frame_ids
    pipelined = use_ids and not args.no_pipeline and cfg.cache_windows
    want_dropin = not args.no_dropin and (use_ids or pipelined)
    # headline call mode: frame groups (forward_group: the backward branches of G consecutive output frames as multi-map launches)
    # where the engine has the multi-map launch list (mid_channels = 24), else one forward() per frame
    G = max(1, min(4, args.group)) if (pipelined and eng.group_ok()) else 1
    if H * W_ > 4 * 270 * 480 and '--group' not in sys.argv:
        G = 1                                       # (1080p -> 8K: four frames' 8K intermediates; one frame per call unless asked for)
    want_percall = G > 1 and not args.no_dropin
    # ---- warm-up proper (the first timed pass used to be the first 36 ms of GPU work of the process): untimed
    # passes of both call modes until >= args.warm_seconds of real work have run -- kernels loaded, allocator pools grown for
    # BOTH modes, clocks and power state settled
    tw = time.perf_counter()
    nwarm = 0
    while True:
        timed_pass(use_ids, pipelined, False, timed=False, group=G)
        if want_percall:
            timed_pass(use_ids, pipelined, False, timed=False)
        if want_dropin:
            timed_pass(False, False, False, timed=False)
        nwarm += 1
        more = time.perf_counter() - tw < args.warm_seconds
        if world > 1:                      # the ranks must agree on the number of passes (they hold barriers): ONE decision, reduced
            flag = torch.tensor([1.0 if more else 0.0], device=dev if backend == 'nccl' else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            more = float(flag.item()) != 0.0
        if not more:
            break
    warm_s = time.perf_counter() - tw
    # the collector must not stop the host inside a 100 ms timed pass: everything allocated so far (model, windows, warm pools)
    # goes to the permanent generation, the young generations are collected between the passes
    import gc
    gc.collect()
    if not args.no_gc_freeze:
        gc.freeze()
    # ---- R timed repetitions, the two call modes interleaved (fast / reference surface / fast / ...): every sample is printed,
    # `value` is the MEDIAN of the fast mode's samples
    samples, samples_dropin, samples_percall, sums, ev = [], [], [], [], None
    for rep in range(max(1, args.repeats)):
        gc.collect()
        el, e = timed_pass(use_ids, pipelined, True, group=G)
        samples.append(el)
        sums.append(stream_summary(e[2], el))
        if ev is None or el <= min(samples):
            ev = e
        if want_percall:
            el1, _ = timed_pass(use_ids, pipelined, False)
            samples_percall.append(el1)
        if want_dropin:
            el2, _ = timed_pass(False, False, False)
            samples_dropin.append(el2)
    elapsed = med(samples)
    fps_samples = [world * args.steps / x for x in samples]
    # (round 6, ADVICE r5: the mode is fixed BEFORE the measurement -- frame groups wherever the engine has the multi-map launch list;
    #  rounds 4-5 reported the faster of the two modes measured, a max-of-medians with a small upward bias.  The other mode's rate stays
    #  in the line as `one_frame_per_call`.)
    headline_mode = 'frame groups' if G > 1 else 'one frame per call'
    # The fused-ResBlock launches are ~10 us each: with the internal streams feeding the GPU concurrently the HIP events around a
    # run of them also bracket the other stream's kernels that get scheduled in between (measured 18.6 us per launch where
    # rocprofv3 reports 9.7).  Their live per-launch time therefore comes from one more pass of the same steps on ONE stream
    # (no cross-call pipelining, no side stream): nothing else is in flight between a run's two events.
    # Group mode (round 5): the launches of record are the MULTI-MAP launches of the backward branches (G maps behind one launch);
    # they are timed in one more pass of the same group calls with every internal section on ONE stream (pipe layout 'one').
    ev_rb = None
    if eng.rb24 and cfg.mid_channels == 24:            # (every rank: timed_pass holds barriers)
        if G > 1:
            lay = getattr(cfg, 'pipe_layout', None)
            cfg.pipe_layout, eng._pipe = 'one', None
            try:
                _, (_, ev_rb, _) = timed_pass(use_ids, pipelined, True, collect_chain=True, group=G)
            finally:
                cfg.pipe_layout, eng._pipe = lay, None
        else:
            ov = eng.overlap
            eng.overlap = False
            try:
                _, (_, ev_rb, _) = timed_pass(use_ids, False, True, collect_chain=True)
            finally:
                eng.overlap = ov
    ev = (ev[0], ev_rb)
    if pipelined:                                      # (the layout the timed passes ran on, not the measurement pass's)
        eng._pipe_streams(dev)
    percall = None
    if samples_percall:
        swapped = headline_mode.startswith('one frame per call (')
        el1 = med(samples_percall)
        pf_ = [world * args.steps / x for x in samples_percall]
        percall = {'value': world * args.steps / el1, 'unit': 'frames/s', 'ms_per_step': 1e3 * el1 / args.steps, 'samples': [round(v, 2) for v in pf_],
                   'call': ('forward_group, %d frames per call (the headline is the one-frame-per-call mode in this run)' % G) if swapped else
                           'one forward(frame_ids=, input_ready=\'materialised\') per output frame, pipelined over the internal streams '
                           '(the round-4 headline mode): one frame of latency per call'}
    dropin = None
    if samples_dropin:
        el2 = med(samples_dropin)
        dfps = [world * args.steps / x for x in samples_dropin]
        dropin = {'value': world * args.steps / el2, 'unit': 'frames/s', 'ms_per_step': 1e3 * el2 / args.steps,
                  'samples': [round(v, 2) for v in dfps], 'min': min(dfps), 'median': world * args.steps / el2, 'max': max(dfps),
                  'call': "net(x, ref, is_first_frame)['result'] -- the reference's positional call, frames recognised by content "
                          '(one device->host flag read per call), default stream order'}

    # first-frame latency (SURVEY 8(d) asks for it beside the steady-state rate): a new clip on a cold window cache,
    # measured AFTER the timed region, host clock around one synchronised call; never part of `value`
    first_ms = None
    if rank == 0:
        try:
            net.Network.reset()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            net(win_lr[0], win_rf[0], True)
            torch.cuda.synchronize()
            first_ms = 1e3 * (time.perf_counter() - t1)
        except Exception:  # noqa: BLE001  (an extra figure must never take the headline number down)
            first_ms = None

    # PCIe-inclusive rate (never `value`): the same K steps with the windows handed over as HOST buffers, the way run.py / eval.py
    # hand them to the network (pinned memory; each call's windows go host -> device on a copy stream ahead of the call, which
    # waits for them through input_ready=<event>; every result goes device -> host behind the call), host clock to the last
    # result in host memory.  One untimed + three timed passes after the timed region.
    pcie = None
    if rank == 0 and world == 1 and pipelined and not args.no_dropin:
        try:
            pcie = pcie_inclusive_pass(net, args, all_lr, all_rf, wins, start, G, dev)
            # ... and with config.result_dtype = 'uint8' (round 6): the output head stores the 8-bit frame itself
            u8 = pcie_inclusive_pass(net, args, all_lr, all_rf, wins, start, G, dev, result_dtype='uint8')
            pcie['result_uint8'] = {k: u8[k] for k in ('value', 'unit', 'samples', 'd2h_mb_per_frame')}
            # ... and with every frame crossing PCIe once (a streaming loader: device-side frame table, windows gathered on the device)
            u8o = pcie_inclusive_pass(net, args, all_lr, all_rf, wins, start, G, dev, result_dtype='uint8', frames_once=True)
            pcie['result_uint8_frames_once'] = {k: u8o[k] for k in ('value', 'unit', 'samples', 'd2h_mb_per_frame')}
            pcie['result_uint8_frames_once']['h2d_mb_per_frame'] = round(2 * all_lr[0, 0].numel() * 4 / 1e6, 2)
        except Exception as e:  # noqa: BLE001  (an extra figure must never take the headline number down)
            pcie = {'error': repr(e)[:300]}
        net.Network.set_pipelined(False)

    # `traffic` of the two roofline kernels measured in THIS run (round 6): two rocprofv3 --pmc child passes (counters only, the GPU is
    # idle meanwhile); any failure -- no rocprofv3, a refused counter, a timeout -- falls back to the committed profiles/pmc_*.json
    live_pmc, live_pmc_err = None, None
    if rank == 0 and world == 1 and not args.no_live_pmc and (H, W_) == (270, 480) and args.config == 'config_RefVSR_small_L1':
        try:
            torch.cuda.synchronize()
            live_pmc = live_pmc_traffic()
        except Exception as e:  # noqa: BLE001  (an extra figure must never take the headline number down)
            live_pmc, live_pmc_err = None, repr(e)[:200]

    line = None
    if rank == 0:
        fps = world * args.steps / elapsed
        alg, parts = tflop_per_frame(cfg, H, W_, T, dedup=True)
        alg_exec, _ = tflop_per_frame(cfg, H, W_, T, dedup=False)
        tag = BASELINE_CONFIG.get(args.config)
        line = {
            'metric': '4x SR frames/sec (%dp->%dp, %s)' % (H, 4 * H, args.config.replace('config_', '')), 'value': fps, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': '%s 4x SR, %dx%d -> %dx%d, frame_num=%d, steady-state sliding window, n=1%s; synthetic clip seed 0, '
                                   'seeded random weights 1234' % (args.config, H, W_, 4 * H, 4 * W_, T, ' (BASELINE %s)' % tag if tag else ''),
                       'frames_per_rank': args.steps, 'parallelism': 'frame-shard x%d (reset-aligned, no collective)' % world,
                       'window_cache': bool(cfg.cache_windows), 'frame_ids': bool(use_ids), 'pipelined_calls': bool(pipelined),
                       'frames_per_call': G, 'headline_mode': headline_mode,
                       'call_surface': (("extended: forward_group(%d consecutive windows, frame_ids) + set_pipelined(True) + input_ready='materialised'" % G)
                                        if G > 1 else "extended: frame_ids= + set_pipelined(True) + input_ready='materialised'" if pipelined else
                                        'extended: frame_ids=' if use_ids else 'reference call surface'),
                       'dropin_frames_per_s': round(dropin['value'], 2) if dropin else None,
                       'one_frame_per_call_frames_per_s': round(percall['value'], 2) if percall else None,
                       'pipe_layout': getattr(eng, 'pipe_layout', None) if pipelined else None,
                       'precision': 'fp16 HWC feature maps + fp16 hi+lo MFMA weights, fp32 accumulate; fp32 matching features / flows / '
                                    'output; arg-max decided at fp32 accuracy (fp16 GEMM top-2 + fp32 re-rank + split-fp16 search of ambiguous columns)'},
            'dropin_surface': dropin,
            'one_frame_per_call': percall,
            # every timed repetition of the fast mode (frames/s); `value` = their median.  The K steps are timed `repeats` times,
            # interleaved with the reference-surface passes, after `warm_seconds` of untimed passes of both modes.
            'samples': [round(v, 2) for v in fps_samples], 'min': min(fps_samples), 'median': fps, 'max': max(fps_samples),
            'repeats': len(fps_samples), 'untimed_warm_passes': nwarm, 'untimed_warm_seconds': round(warm_s, 3),
            # per-stream section times of the median-nearest and of the slowest pass (HIP events around the P / F / M sections of
            # every call): a slow mode shows up HERE as stretched sections, not only as a lower rate
            'streams': {'median_pass': sums[min(range(len(samples)), key=lambda i: abs(samples[i] - elapsed))],
                        'slowest_pass': sums[max(range(len(samples)), key=lambda i: samples[i])],
                        'slow_passes': [{'rep': i, 'frames_per_s': round(fps_samples[i], 2), 'sections': sums[i]}
                                        for i in range(len(samples)) if fps_samples[i] < 0.9 * max(fps_samples)]},
        }
        # ---- rooflines from HIP events.  `roofline` = the time-dominant kernel: the fused 24-channel ResBlock
        # (resblock24_kernel, 156 launches per frame, ~28 % of the device time; events bracket every run of >= 8 blocks on the
        # LR map in the single-stream pass above, duration / blocks = per-launch time incl. the gaps between the launches);
        # `roofline_match_top2` = the matching GEMM (one launch per frame, ~14 %).  Configurations whose blocks do not run on
        # that kernel (C = 48) report the matching kernel as `roofline`.
        ev, cev = ev if ev else (None, None)
        rb_line = None
        # chain events: (start, end, blocks, h, w[, maps per launch]); group mode: the multi-map launches only
        cev5 = [(c[0], c[1], c[2], c[3], c[4], (c[5] if len(c) > 5 else 1)) for c in (cev or [])]
        maps_per_launch = G if any(c[5] == G for c in cev5) else 1
        runs = [(a.elapsed_time(b), n) for a, b, n, hh, ww, bb in cev5 if (hh, ww) == (H, W_) and n >= 8 and bb == maps_per_launch]
        if runs:
            per_launch_ms = sum(m for m, _ in runs) / sum(n for _, n in runs)
            C_ = cfg.mid_channels
            flops = maps_per_launch * 2 * 2.0 * 9 * C_ * C_ * H * W_
            ach = flops / (per_launch_ms * 1e-3) / 1e12
            traffic, tsrc = None, None
            pj = os.path.join(ROOT, 'profiles', 'pmc_kernels.json')
            if live_pmc is not None and (H, W_) == (270, 480):
                tj = live_pmc[0].get('traffic_bytes_per_launch', {})
                traffic = tj.get('resblock LR x%d maps' % maps_per_launch) if maps_per_launch > 1 else tj.get('resblock LR')
                tsrc = ('LIVE: two rocprofv3 --pmc child passes of this run (FETCH_SIZE, WRITE_SIZE: separate processes, counters only) on '
                        'tools/pmc_kernels.py; FETCH_SIZE x 2 (gfx950, 16-byte coalesced reads) + WRITE_SIZE, KiB units')
            if traffic is None and os.path.exists(pj) and (H, W_) == (270, 480):
                try:
                    tj = json.load(open(pj)).get('traffic_bytes_per_launch', {})
                    traffic = tj.get('resblock LR x%d maps' % maps_per_launch) if maps_per_launch > 1 else tj.get('resblock LR')
                    if traffic is None and maps_per_launch > 1 and tj.get('resblock LR'):
                        traffic = tj['resblock LR'] * maps_per_launch
                    tsrc = 'STATIC: profiles/pmc_kernels.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE in separate passes on tools/pmc_kernels.py; re-measured on the round-6 tree, tools/gpu_runs/r6_call8.sh, raw counters profiles/r06_pmc_kernels_*.csv), not measured in this run'
                except Exception:  # noqa: BLE001
                    traffic = None
            # Both roofs.  Algorithmic bytes per launch = every map read once + written once (fp16 HWC, 48 bytes per
            # pixel each way) + one weight blob; arithmetic intensity = useful FLOPs / those bytes; the roof that GOVERNS is the HBM one
            # when the intensity lies below the ridge (peak FLOP/s / peak bytes/s = 312 FLOP/byte), else the MFMA one.  `achieved` /
            # `peak` / `frac` are those of the governing roof, the other roof's fraction is beside it.
            abytes = maps_per_launch * 2.0 * H * W_ * 48 + 43264
            ai = flops / abytes
            ridge = PEAK_F16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            hbm_gbs = abytes / (per_launch_ms * 1e-3) / 1e9
            governing = 'hbm' if ai < ridge else 'mfma'
            issued = 798.0 * 16384 / (2 * 2.0 * 9 * C_ * C_ * 256)
            rb_line = {'kernel': 'resblock24_kernel (fused conv3x3-ReLU-conv3x3+residual, 24 channels, LR map %dx%d%s)' %
                                 (H, W_, ', %d maps per launch' % maps_per_launch if maps_per_launch > 1 else ''), 'bound': governing,
                       'maps_per_launch': maps_per_launch,
                       'achieved': hbm_gbs if governing == 'hbm' else ach, 'peak': PEAK_HBM_GBS if governing == 'hbm' else PEAK_F16_TFLOPS,
                       'unit': 'GB/s' if governing == 'hbm' else 'TFLOP/s',
                       'frac': (hbm_gbs / PEAK_HBM_GBS) if governing == 'hbm' else ach / PEAK_F16_TFLOPS,
                       'governing': '%s (arithmetic intensity %.0f FLOP/byte %s the ridge %.0f)' % (governing, ai, '<' if ai < ridge else '>=', ridge),
                       'mfma_frac': ach / PEAK_F16_TFLOPS, 'mfma_tflops': ach, 'mfma_issued_frac': issued * ach / PEAK_F16_TFLOPS,
                       'hbm_frac': hbm_gbs / PEAK_HBM_GBS, 'hbm_gbs': hbm_gbs, 'algorithmic_bytes_per_launch': abytes, 'arithmetic_intensity': ai,
                       'traffic': traffic, 'traffic_source': tsrc, 'traffic_static': None if traffic is None else not str(tsrc).startswith('LIVE'),
                       'traffic_over_algorithmic': None if traffic is None else traffic / abytes,
                       'launches_timed': sum(n for _, n in runs),
                       'launches_timed_in': 'one more pass of the SAME calls with every internal section on one stream (in the timed passes the '
                                            'events around a run also bracket the other streams\' kernels)',
                       'mean_launch_ms': per_launch_ms, 'flops_per_launch': flops,
                       'issued_over_useful_flops': 798.0 * 16384 / (2 * 2.0 * 9 * C_ * C_ * 256),
                       'note': 'useful FLOPs (2 x 9 x 24 x 24 x 2 convs per pixel); the kernel issues 2.46x that on the matrix pipe: x2 hi + lo '
                               'weight halves (48 rows in 3 fragments), x1.19 the 10 x 34 halo region of conv1, x1.04 K = 216 padded to 224'}
        ms = [a.elapsed_time(b) for a, b in (ev or [])]
        if ms:
            mean_ms = sum(ms) / len(ms)
            hm, wm = (H // (cfg.scale // 2), W_ // (cfg.scale // 2)) if cfg.flag_HD_in else (H, W_)
            n_lr, n_ref = ((hm // 2) * (wm // 2), (hm // 4) * (wm // 4)) if cfg.flag_HD_in else (hm * wm, (hm // 2) * (wm // 2))
            flops = 2.0 * n_lr * n_ref * 144
            ach = flops / (mean_ms * 1e-3) / 1e12
            traffic, tsrc = None, None
            pj = os.path.join(ROOT, 'profiles', 'pmc_match_top2.json')
            if live_pmc is not None and (n_lr, n_ref) == (129600, 32400):
                traffic = live_pmc[1].get('traffic_bytes_per_launch')
                tsrc = 'LIVE: two rocprofv3 --pmc child passes of this run (FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE)'
            if traffic is None and os.path.exists(pj) and (n_lr, n_ref) == (129600, 32400):   # HBM bytes per launch from the committed PMC passes
                try:
                    traffic = json.load(open(pj))['traffic_bytes_per_launch']
                    tsrc = 'STATIC: profiles/pmc_match_top2.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; re-measured on the round-6 tree, tools/gpu_runs/r6_call8.sh), not measured in this run'
                except Exception:  # noqa: BLE001
                    traffic = None
            mbytes = (n_lr + n_ref) * 304.0 + n_lr * 16.0            # fp16 rows of 152 halfs, read once; top-2 (index, value) written
            line['roofline_match_top2'] = {'kernel': 'match_top2_kernel (fused cosine GEMM + column top-2)', 'bound': 'mfma',
                                           'achieved': ach, 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F16_TFLOPS,
                                           'governing': 'mfma (arithmetic intensity %.0f FLOP/byte >> the ridge %.0f)' % (flops / mbytes, PEAK_F16_TFLOPS * 1e3 / PEAK_HBM_GBS),
                                           'mfma_frac': ach / PEAK_F16_TFLOPS, 'hbm_frac': mbytes / (mean_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                           'traffic': traffic, 'traffic_source': tsrc, 'traffic_static': None if traffic is None else not str(tsrc).startswith('LIVE'),
                                           'launches_timed': len(ms), 'launches_timed_in': 'the timed passes themselves (one launch per frame, events around it)',
                                           'mean_launch_ms': mean_ms, 'flops_per_launch': flops}
        else:
            line['roofline_match_top2'] = None
        line['roofline'] = rb_line if rb_line is not None else line['roofline_match_top2']
        if live_pmc_err and isinstance(line['roofline'], dict):
            line['roofline']['live_pmc_error'] = live_pmc_err
        line['first_frame_ms'] = first_ms
        line['pcie_inclusive'] = pcie
        sv = SURVEY_DEDUP_TFLOP.get(args.config) if (H, W_, T) == (270, 480, 5) else None
        line['whole_path'] = {
            'algorithmic_tflop_per_frame': alg, 'breakdown': {k: round(v, 4) for k, v in parts.items()},
            'counting': 'refvsr_amd/flops.py: every conv / GEMM of the layer list once per output frame (2 SPyNet, 1 matching, 1 encoder + '
                        'alignment pass, %d+1 propagation steps, 1 upsampler); as the reference executes a call: %.3f' % (T - T // 2, alg_exec),
            'achieved_tflops_per_gpu': alg * fps / world, 'frac_of_f16_mfma_peak': alg * fps / world / PEAK_F16_TFLOPS,
            'survey_8d_tflop_per_frame': sv,
            'frac_of_f16_mfma_peak_on_survey_figure': (sv * fps / world / PEAK_F16_TFLOPS) if sv else None}
        if not args.no_kernels:
            try:
                line['kernels'] = kernel_rooflines(cfg, eng, H, W_, dev, None if live_pmc is None else live_pmc[0].get('traffic_bytes_per_launch'))
            except Exception as e:  # noqa: BLE001
                line['kernels'] = {'error': repr(e)[:300]}
        if world == 1 and not args.no_wavefront and (H, W_) == (270, 480):
            try:
                line['wavefront_model'] = wavefront_model_single_gpu(args, dev, H, W_)
            except Exception as e:  # noqa: BLE001
                line['wavefront_model'] = {'error': repr(e)[:300]}
        if world == 1 and not args.no_other_configs and args.config == 'config_RefVSR_small_L1' and (H, W_) == (270, 480):
            # BASELINE configs[2] and configs[4] on this GPU, after the headline: never in `value`
            del win_lr, win_rf, lr, rf
            net.Network.reset()
            torch.cuda.empty_cache()
            line['other_configs'] = {}
            for key, (nm, hh, ww, st, wu) in (('configs[2]', ('config_RefVSR_MFID', 270, 480, 12, 3)),
                                              ('configs[4] on one GPU', ('config_RefVSR_MFID_8K', 1080, 1920, 4, 2))):
                try:
                    torch.cuda.reset_peak_memory_stats(dev)
                    line['other_configs'][key] = other_config_leg(nm, hh, ww, st, wu, dev)
                except Exception as e:  # noqa: BLE001
                    line['other_configs'][key] = {'error': repr(e)[:300]}
    if dist_on and not args.no_wavefront:
        # The extra leg must never take the headline number down: if it has not returned within the deadline (a hung
        # send / recv, a rank that died) every rank leaves through a watchdog, rank 0 after printing the line.
        import threading

        def bail():
            if rank == 0:
                promote_wavefront(line, {'error': 'not finished within %.0f s' % args.wavefront_timeout}, args, world)
                line['cpu_baseline'] = None
                emit(line, args)
            os._exit(0)
        dog = threading.Timer(args.wavefront_timeout, bail)
        dog.daemon = True
        dog.start()
        try:
            wf = run_wavefront_leg(args, rank, world, dev, backend, H, W_)
        except Exception as e:  # noqa: BLE001
            wf = {'error': repr(e)[:400]}
            if rank == 0:
                promote_wavefront(line, wf, args, world)
                line['cpu_baseline'] = None
                emit(line, args)
            os._exit(0)                       # the other ranks may be blocked in a collective: do not wait for them
        dog.cancel()
        if rank == 0:
            promote_wavefront(line, wf, args, world)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            import subprocess
            try:                     # child process + timeout: the baseline leg must never take the GPU number down
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--config', args.config,
                                    '--size', args.size, '--frames', str(T)],
                                   capture_output=True, text=True, timeout=args.cpu_baseline_timeout)
                tagl = [ln for ln in r.stdout.splitlines() if ln.startswith('CPU_BASELINE ')]
                if not tagl:
                    raise RuntimeError('no result (rc=%d): %s' % (r.returncode, r.stderr[-300:]))
                line['cpu_baseline'] = json.loads(tagl[-1][len('CPU_BASELINE '):])
                line['cpu_baseline']['gpu_over_cpu'] = line['value'] / line['cpu_baseline']['value']
            except Exception as e:  # noqa: BLE001
                line['cpu_baseline'] = {'error': repr(e)[:300]}
        else:
            line['cpu_baseline'] = None
        emit(line, args)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
