#!/usr/bin/env python3
"""Headline benchmark: 4x SR output frames/s of RefVSR_small (270x480 -> 1080x1920, frame_num=5),
steady-state sliding-window inference through the drop-in SRNet surface on the HIP path.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one forward call = one 1080p output frame.  Inputs (the synthetic LR / Ref clip) are
resident in HBM before the timed region.  With N > 1 every rank runs K steps on its own
reset-aligned shard of one long clip (exchange-free partition, refvsr_amd/shard.py), so per-GPU
work is fixed ("weak") and there is no data-path collective; the timed region is bracketed by
barrier + synchronize and the MAX over ranks is reported.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (fused matching GEMM + arg-max, MFMA-bound): algorithmic
                  FLOPs per launch / mean launch duration measured with HIP events on the launch
                  stream during the timed steps, against the 2.5 PFLOP/s dense fp16 MFMA peak.
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm; the reference itself cannot
                  travel) timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, W_, T = 270, 480, 5
PEAK_F16_TFLOPS = 2500.0          # dense MFMA f16/bf16 peak, MI355X_MICROARCH.md
ALG_TFLOP_PER_FRAME = 2.490       # de-duplicated algorithmic work per output frame


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


def cpu_baseline_child():
    """Runs in a child process (so a slow host can be cut off without losing the GPU number)."""
    from refvsr_amd import get_config, make_state_dict
    from refvsr_amd.synth import make_clip
    ncores = min(usable_cores(), 64)
    torch.set_num_threads(ncores)
    cfg = get_config('bench', 'bench', 'config_RefVSR_small_L1')
    cfg.frame_num = T
    sd = make_state_dict(cfg, 1234)
    lr, rf, _ = make_clip(T, H, W_, seed=0)
    out = cpu_baseline(cfg, sd, lr, rf)
    out['host_cpu_count'] = os.cpu_count()
    out['usable_cores'] = usable_cores()
    print('CPU_BASELINE ' + json.dumps(out), flush=True)


def cpu_baseline(cfg, sd, lr, rf):
    """One steady-state forward of the oracle exactly as the reference executes it (8 SPyNet calls,
    3 matchings, 3 backward + 1 forward RAP steps, upsampler) at the full 270x480 size.  The
    matching GEMM (86% of the reference's CPU time) is evaluated on 1/16 of the LR columns and its
    time scaled back by 16; everything else runs in full."""
    from oracle import refvsr_oracle as orc
    nthreads = torch.get_num_threads()
    sample = 16
    o = orc.OracleNetwork(cfg, sd, match_chunk=8192, match_sample=sample)
    # forward state of a previous call (values do not influence the timing)
    C = cfg.mid_channels
    o.forward_feat_prop_prev = torch.zeros(1, C, H, W_)
    o.forward_flow_prev = torch.zeros(1, 2, H, W_)
    o.forward_feat_prop_UP_prev = torch.zeros(1, C, 2 * H, 2 * W_)
    o.forward_conf_map_prop_prev = torch.zeros(1, 1, H, W_)
    o.frame_itr_num = 1
    x, r = lr[:T][None].cpu(), rf[:T][None].cpu()
    with torch.no_grad():
        t0 = time.perf_counter()
        o.forward(x, r, False)
        total = time.perf_counter() - t0
    scaled = (total - o.match_seconds) + o.match_seconds * sample
    return {
        'value': 1.0 / scaled, 'unit': 'frames/s', 'cores': nthreads, 'kind': 'port',
        'sample': ('1 steady-state forward as the reference executes it (8 SPyNet, 3 matchings, 3+1 RAP steps, '
                   'upsampler) at 270x480 t=5 fp32, torch CPU; matching GEMM on 1/%d of the LR columns, '
                   'its time (%.2f s) scaled x%d; measured %.2f s -> %.2f s/frame' %
                   (sample, o.match_seconds, sample, total, scaled)),
        'seconds_per_frame': scaled,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', default='config_RefVSR_small_L1')
    ap.add_argument('--no-cache', action='store_true', help='execute exactly the work the reference executes')
    ap.add_argument('--no-frame-ids', action='store_true', help='let the engine recognise frames by content comparison')
    ap.add_argument('--no-pipeline', action='store_true', help='do not overlap consecutive calls on internal streams')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline-timeout', type=float, default=240.0)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_child()
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d needs a torch.distributed.run launch with --nproc-per-node %d' % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    # REFVSR_DIST_BACKEND=gloo lets the N>1 code path be exercised on a single-GPU box (all ranks share GPU 0);
    # the real multi-GPU run uses nccl (= RCCL), one rank per GPU.
    backend = os.environ.get('REFVSR_DIST_BACKEND', 'nccl')
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world)

    from refvsr_amd import SRNet, get_config, make_state_dict
    from refvsr_amd.synth import make_clip, window_indices
    cfg = get_config('bench', 'bench', args.config)
    cfg.frame_num = T
    cfg.cache_windows = not args.no_cache
    sd = make_state_dict(cfg, 1234)
    net = SRNet(cfg).to(dev).eval()
    net.load_state_dict(sd)

    nfr = args.warmup + args.steps
    R = cfg.reset_branch or nfr
    start = rank * int(math.ceil(nfr / float(R))) * R          # reset-aligned shard start (exchange-free)
    lr, rf, _ = make_clip(nfr, H, W_, seed=0, start=start)
    lr, rf = lr.to(dev), rf.to(dev)                             # inputs resident in HBM
    # the sliding windows are materialised before the timed region (inputs resident in HBM); frame ids let the engine
    # key its window cache without comparing frame contents and pipeline consecutive calls over its internal streams
    wins = [window_indices(f, nfr, T) for f in range(nfr)]
    win_lr = [lr[torch.tensor(w, device=dev)][None].contiguous() for w in wins]
    win_rf = [rf[torch.tensor(w, device=dev)][None].contiguous() for w in wins]
    use_ids = not args.no_frame_ids
    if use_ids and not args.no_pipeline:
        net.Network.set_pipelined(True)
    torch.cuda.synchronize()

    def step(f):
        ids = [start + i for i in wins[f]] if use_ids else None
        return net(win_lr[f], win_rf[f], f == 0, frame_ids=ids)['result']

    eng = net.Network.ensure_engines(1, dev)[0]
    for f in range(args.warmup):
        out = step(f)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.kernel_events = []                                       # HIP events around the dominant kernel
    t0 = time.perf_counter()
    for f in range(args.warmup, nfr):
        out = step(f)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(out).all())

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

    if rank == 0:
        fps = world * args.steps / elapsed
        line = {
            'metric': '4x SR frames/sec (270p->1080p, RefVSR_small)', 'value': fps, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': '%s 4x SR, 270x480 -> 1080x1920, frame_num=5, steady-state sliding window, n=1 '
                                   '(BASELINE configs[1]); synthetic clip seed 0, seeded random weights 1234' % args.config,
                       'frames_per_rank': args.steps, 'parallelism': 'frame-shard x%d (reset-aligned, no collective)' % world,
                       'window_cache': bool(cfg.cache_windows), 'frame_ids': use_ids,
                       'pipelined_calls': bool(use_ids and not args.no_pipeline and cfg.cache_windows),
                       'precision': 'fp16 HWC feature maps + fp16 MFMA operands, fp32 accumulate; fp32 matching features / flows / output'},
        }
        # ---- roofline of the dominant kernel (match_top2) from the events recorded in the timed region
        ev = getattr(eng, 'kernel_events', None) or []
        if ev:
            ms = [a.elapsed_time(b) for a, b in ev]
            mean_ms = sum(ms) / len(ms)
            n_lr, n_ref = H * W_, (H // 2) * (W_ // 2)
            flops = 2.0 * n_lr * n_ref * 144
            ach = flops / (mean_ms * 1e-3) / 1e12
            traffic, tsrc = None, None
            pj = os.path.join(ROOT, 'profiles', 'pmc_match_top2.json')
            if os.path.exists(pj):           # HBM bytes per launch from the PMC passes (tools/pmc_to_json.py); not live
                try:
                    traffic = json.load(open(pj))['traffic_bytes_per_launch']
                    tsrc = 'profiles/pmc_match_top2.json (rocprofv3 --pmc FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)'
                except Exception:  # noqa: BLE001
                    traffic = None
            line['roofline'] = {'kernel': 'match_top2_kernel (fused cosine GEMM + column top-2)', 'bound': 'mfma',
                                'achieved': ach, 'peak': PEAK_F16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F16_TFLOPS,
                                'traffic': traffic, 'traffic_source': tsrc, 'launches_timed': len(ms),
                                'mean_launch_ms': mean_ms, 'flops_per_launch': flops}
        else:
            line['roofline'] = None
        line['first_frame_ms'] = first_ms
        line['whole_path'] = {'algorithmic_tflop_per_frame_dedup': ALG_TFLOP_PER_FRAME,
                              'achieved_tflops_per_gpu': ALG_TFLOP_PER_FRAME * fps / world,
                              'frac_of_f16_mfma_peak': ALG_TFLOP_PER_FRAME * fps / world / PEAK_F16_TFLOPS}
        if world == 1 and not args.no_cpu_baseline:
            import subprocess
            try:                     # child process + timeout: the baseline leg must never take the GPU number down
                r = subprocess.run([sys.executable, os.path.abspath(__file__), '--cpu-baseline-only'],
                                   capture_output=True, text=True, timeout=args.cpu_baseline_timeout)
                tag = [ln for ln in r.stdout.splitlines() if ln.startswith('CPU_BASELINE ')]
                if not tag:
                    raise RuntimeError('no result (rc=%d): %s' % (r.returncode, r.stderr[-300:]))
                line['cpu_baseline'] = json.loads(tag[-1][len('CPU_BASELINE '):])
                line['cpu_baseline']['gpu_over_cpu'] = fps / line['cpu_baseline']['value']
            except Exception as e:  # noqa: BLE001
                line['cpu_baseline'] = {'error': repr(e)[:300]}
        else:
            line['cpu_baseline'] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
