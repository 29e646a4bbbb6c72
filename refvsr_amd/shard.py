"""Multi-GPU execution of one clip: shard by output-frame index, one process per GPU.

The only cross-frame dependency of the path is the forward-branch state
(forward_feat_prop_prev, forward_flow_prev, forward_feat_prop_UP_prev, forward_conf_map_prop_prev,
frame_itr_num -- RefVSR.py:96-101,279-283).  The backward branch restarts from zeros in every window
(:211-214), and with `reset_branch = R` call k*R behaves as is_first_frame=True (:168-170).  Hence

  * shards that start at a multiple of R are bit-identical to the sequential run with NO
    communication -- `partition(..., aligned=True)`;
  * any other shard boundary needs ONE point-to-point hand-off of the state from rank r to r+1
    (RCCL send/recv over a single xGMI link; 63.8 MB fp32 for RefVSR_small at 270p).  There is no
    all-reduce anywhere on the inference path.

  * the hand-off chain serialises the ranks if each waits for the previous one to finish its whole
    shard (`run_sharded`).  `run_wavefront` splits every frame into phase A -- flows, matching,
    reference encoders, alignment and the whole backward branch, 85-90 % of the work, independent of
    the carried state -- and phase B -- one forward-branch step + the upsampler.  All ranks run phase A
    of all their frames concurrently; phase B then runs as a wavefront rank 0 -> 1 -> ... behind the
    hand-off, so the serial part of an N-rank run is ~(10-15 %) x nframes instead of 100 %.

`run_sharded` / `run_wavefront` are executor-agnostic (anything with forward or phase_a/phase_b and
export_state/import_state), so the protocols are tested on CPU with the gloo backend and the oracle as
executor, and on one GPU with two processes and the real engine.
"""
import torch
import torch.distributed as dist

STATE_KEYS = ('feat', 'flow', 'feat_up', 'conf')
MAX_KEYFRAMES = 11


def partition(nframes, world, reset_branch=None, aligned=False):
    """Contiguous [start, end) frame ranges per rank.  aligned=True snaps boundaries to multiples of
    reset_branch (exchange-free, possibly unbalanced)."""
    if aligned:
        assert reset_branch, 'aligned partition needs reset_branch'
        units = list(range(0, nframes, reset_branch)) + [nframes]
        nunits = len(units) - 1
        per = [nunits // world + (1 if r < nunits % world else 0) for r in range(world)]
        out, u = [], 0
        for r in range(world):
            out.append((units[u], units[u + per[r]]))
            u += per[r]
        return out
    base, rem = divmod(nframes, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def needs_handoff(start, reset_branch):
    return start != 0 and not (reset_branch and start % reset_branch == 0)


def send_state(state, dst, device, async_op=False):
    """Hand the forward-branch state to rank `dst`.

    * `state` a tensor: the packed device buffer of Engine.export_state_packed() (fp16 HWC maps + fp32 flow / conf in
      ONE message, 32.7 MB for RefVSR_small at 270p) -- the RCCL path: one send, no unpack kernels; `async_op=True`
      returns the work handle of an isend so that the sender's remaining kernels (the upsampler of its last frame) run
      under the transfer.
    * `state` a dict of planar fp32 tensors + frame_itr_num (executor-agnostic form, e.g. the oracle on gloo)."""
    if torch.is_tensor(state):
        buf = state if str(state.device).startswith(str(device)) else state.to(device)
        if async_op:
            return dist.isend(buf, dst)
        dist.send(buf, dst)
        return None
    kf = [float(k) for k in (state.get('keyframe_idx') or [])]          # RefVSR_IR: key-frame indices travel with the state
    assert len(kf) <= MAX_KEYFRAMES
    meta = torch.tensor([float(state['frame_itr_num'])] + [float(d) for k in STATE_KEYS for d in state[k].shape[-2:]] +
                        [float(len(kf))] + kf + [0.0] * (MAX_KEYFRAMES - len(kf)), dtype=torch.float32, device=device)
    dist.send(meta, dst)
    for k in STATE_KEYS:
        dist.send(state[k].contiguous().to(device), dst)
    return None


def recv_state(src, channels, device, nbytes=None):
    """Counterpart of send_state; nbytes: size of the packed buffer (Engine.state_nbytes) -> returns that buffer."""
    if nbytes is not None:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dist.recv(buf, src)
        return buf
    meta = torch.empty(1 + 2 * len(STATE_KEYS) + 1 + MAX_KEYFRAMES, dtype=torch.float32, device=device)
    dist.recv(meta, src)
    meta = meta.cpu().tolist()
    chans = {'feat': channels, 'flow': 2, 'feat_up': channels, 'conf': 1}
    st = {'frame_itr_num': int(meta[0])}
    nk = int(meta[1 + 2 * len(STATE_KEYS)])
    if nk:
        st['keyframe_idx'] = [int(v) for v in meta[2 + 2 * len(STATE_KEYS):2 + 2 * len(STATE_KEYS) + nk]]
    for i, k in enumerate(STATE_KEYS):
        hh, ww = int(meta[1 + 2 * i]), int(meta[2 + 2 * i])
        buf = torch.empty((chans[k], hh, ww), dtype=torch.float32, device=device)
        dist.recv(buf, src)
        st[k] = buf
    return st


def _import(executor, src, channels, device):
    """Receive the state from rank src into the executor (packed single-message form when the executor offers it)."""
    nb = executor.state_nbytes() if hasattr(executor, 'state_nbytes') else None
    st = recv_state(src, channels, device, nb)
    if nb is not None:
        executor.import_state_packed(st)
    else:
        executor.import_state(st)


def _export(executor):
    return executor.export_state_packed() if hasattr(executor, 'export_state_packed') else executor.export_state()


def run_sharded(executor, get_window, nframes, frame_num, reset_branch, channels, device, aligned=False,
                on_result=None):
    """Run this rank's share of an nframes clip.

    executor(lrs[t,3,h,w], refs, is_first_frame) -> result; executor.export_state()/.import_state(st).
    get_window(f) -> (lrs, refs) of output frame f.  Returns {frame: result} for the local frames."""
    rank, world = dist.get_rank(), dist.get_world_size()
    start, end = partition(nframes, world, reset_branch, aligned)[rank]
    results = {}
    if end > start and needs_handoff(start, reset_branch):
        _import(executor, rank - 1, channels, device)
        first = False
    else:
        first = True
    for f in range(start, end):
        lrs, refs = get_window(f)
        out = executor(lrs, refs, first)
        first = False
        results[f] = out
        if on_result is not None:
            on_result(f, out)
    nxt = partition(nframes, world, reset_branch, aligned)[rank + 1] if rank + 1 < world else None
    if nxt is not None and nxt[1] > nxt[0] and needs_handoff(nxt[0], reset_branch):
        send_state(_export(executor), rank + 1, device)
    return results


def partition_hybrid(nframes, world, reset_branch):
    """The partition the wavefront uses when the forward branch restarts every reset_branch frames: boundaries snapped to
    multiples of reset_branch (those shards need no hand-off and start immediately), except that a short last shard is
    re-balanced with its predecessor -- ONE boundary then lies inside a restart unit and is served by the hand-off.
    64 frames, reset 9, 8 ranks: (0,9) (9,18) ... (45,54) (54,59) (59,64): seven exchange-free starts, one hand-off, the longest
    shard has 9 frames (64 / 9 = 7.1 x over one rank; the balanced partition needs a hand-off at every boundary)."""
    parts = partition(nframes, world, reset_branch, aligned=True)
    nz = [i for i, (a, b) in enumerate(parts) if b > a]
    if len(nz) >= 2:
        i, j = nz[-2], nz[-1]
        (a0, a1), (b0, b1) = parts[i], parts[j]
        if (a1 - a0) - (b1 - b0) >= 2:
            mid = a0 + (b1 - a0 + 1) // 2
            parts[i], parts[j] = (a0, mid), (mid, b1)
    return parts


def partition_chain(nframes, world, ratio=0.165):
    """Partition for a clip WITHOUT forward-branch restarts (reset_branch = None, BASELINE configs[4]): every boundary needs
    the hand-off, so the B1 steps of all frames form one serial chain over the ranks.  A rank can start its chain when its own
    phase A is done AND the state has arrived; equal shards make rank r wait for r whole shards of B1 (4.1 x at 8 ranks with
    the measured phase times).  Shards growing by (1 + ratio) per rank (ratio = t_B1 / t_A, 0.165 measured for RefVSR_small
    at 270p: profiles/r03_bench.json `wavefront_model`) let the chain arrive exactly when phase A ends: n_0 = x,
    n_r = x (1 + ratio)^r.  64 frames over 8 ranks: 4 5 6 7 8 10 11 13 -> 4.8 x predicted (equal shards: 4.1 x)."""
    if nframes < world:
        return partition(nframes, world)
    w = [(1.0 + ratio) ** r for r in range(world)]
    tot = sum(w)
    ideal = [nframes * v / tot for v in w]
    n = [max(1, int(v)) for v in ideal]          # every rank owns at least one frame (nframes >= world)
    # largest remainders first ...
    while sum(n) < nframes:
        r = max(range(world), key=lambda i: (ideal[i] - n[i], i))
        n[r] += 1
    # ... and when the floor-to-one bump overshot: take from the largest shard (ties -> highest rank), never below 1
    while sum(n) > nframes:
        r = max(range(world), key=lambda i: (n[i], i))
        assert n[r] > 1
        n[r] -= 1
    n.sort()                                      # non-decreasing sizes along the chain
    out, s0 = [], 0
    for v in n:
        out.append((s0, s0 + v))
        s0 += v
    return out


def run_wavefront(executor, get_window, nframes, frame_num, reset_branch, channels, device, on_result=None, parts=None,
                  timings=None):
    """Two-phase run of this rank's share (contiguous ranges, any boundary; default: the balanced partition).

    executor.phase_a(lrs, refs, frame_index, first_hint) -> handle   (state-free; all ranks concurrently)
    executor.phase_b1(handle, is_first_frame) -> handle              (forward-branch step: carries the state; in frame order)
    executor.phase_b2(handle) -> result                              (BW/FW fusion + upsampler: state-free)
    Order on every rank: phase A of all local frames | receive the state | B1 of all local frames | send the state | B2 of
    all local frames -- the serial chain over the ranks carries only the B1 steps; the upsamplers of rank r run while rank
    r + 1 walks its B1 chain.  Executors without phase_b1 / phase_b2 run phase_b(handle, first) per frame instead (B2 then
    sits on the chain).  Results are identical to the sequential run.  Returns {frame: result} for the local frames.
    timings (optional dict): filled with host-side seconds of the three phases of this rank (after device synchronisation
    when the executor offers .sync())."""
    import time
    rank, world = dist.get_rank(), dist.get_world_size()
    if parts is None:
        parts = partition(nframes, world)
    start, end = parts[rank]
    sync = getattr(executor, 'sync', lambda: None)
    t0 = time.perf_counter()
    handles = {}
    for f in range(start, end):                                   # ---- phase A: no communication, no state
        lrs, refs = get_window(f)
        hint = f == 0 or bool(reset_branch and f % reset_branch == 0) or (f == start and not needs_handoff(start, reset_branch))
        handles[f] = executor.phase_a(lrs, refs, f, hint)
    if timings is not None:
        sync()
        timings['phase_a'] = time.perf_counter() - t0
    results = {}
    if end > start and needs_handoff(start, reset_branch):        # ---- phase B: wavefront behind the hand-off
        _import(executor, rank - 1, channels, device)
        first = False
    else:
        first = True
    t1 = time.perf_counter()
    nxt = parts[rank + 1] if rank + 1 < world else None
    handoff = nxt is not None and nxt[1] > nxt[0] and needs_handoff(nxt[0], reset_branch)
    pending = []
    if hasattr(executor, 'phase_b1'):
        for f in range(start, end):                               # B1 chain: forward-branch steps only
            handles[f] = executor.phase_b1(handles[f], first)
            first = False
        if handoff:                                               # the next rank's chain starts as soon as this one ends
            pending.append(send_state(_export(executor), rank + 1, device, async_op=True))
        if timings is not None:
            sync()
            timings['phase_b1'] = time.perf_counter() - t1
        t2 = time.perf_counter()
        for f in range(start, end):                               # B2: upsamplers, off the chain
            out = executor.phase_b2(handles.pop(f))
            results[f] = out
            if on_result is not None:
                on_result(f, out)
        if timings is not None:
            sync()
            timings['phase_b2'] = time.perf_counter() - t2
    else:
        def start_send():    # called by phase_b of the LAST local frame as soon as its carried state is final:
            pending.append(send_state(_export(executor), rank + 1, device, async_op=True))   # the upsampler runs under the send
        early = handoff and getattr(executor, 'supports_after_state', False)
        for f in range(start, end):
            if early and f == end - 1:
                out = executor.phase_b(handles.pop(f), first, after_state=start_send)
            else:
                out = executor.phase_b(handles.pop(f), first)
            first = False
            results[f] = out
            if on_result is not None:
                on_result(f, out)
        if handoff and not pending:
            send_state(_export(executor), rank + 1, device)
    for wk in pending:
        if wk is not None:
            wk.wait()
    return results


def predicted_speedup(nframes, world, parts, reset_branch, t_a, t_b1, t_b2, t_handoff=0.0):
    """Makespan model of run_wavefront from per-frame phase times: every rank runs its phase A frames, the B1 chain of a
    shard that starts behind a hand-off begins when its predecessor's chain has ended (+ the hand-off), B2 follows B1 on the
    same rank.  Returns (speedup over one rank, makespan in the same unit as the inputs)."""
    end_b1 = []
    span = 0.0
    for r, (a, b) in enumerate(parts):
        n = b - a
        if n <= 0:
            end_b1.append(end_b1[-1] if end_b1 else 0.0)
            continue
        ready = n * t_a
        if needs_handoff(a, reset_branch) and r > 0:
            ready = max(ready, end_b1[r - 1] + t_handoff)
        e = ready + n * t_b1
        end_b1.append(e)
        span = max(span, e + n * t_b2)
    seq = nframes * (t_a + t_b1 + t_b2)
    return (seq / span if span > 0 else 0.0), span


class EngineExecutor(object):
    """Adapter of the HIP `SRNet` (refvsr_amd/model.py) to run_sharded / run_wavefront: windows are named by their frame
    indices (id-keyed window cache, no content compare), the hand-off uses the packed single-message state."""
    supports_after_state = True

    def __init__(self, net, device, h, w, nframes, frame_num, keep_on_device=True, pipelined=True, inputs_materialised=False):
        self.net, self.dev, self.h, self.w, self.nframes, self.t = net, device, h, w, nframes, frame_num
        self.keep = keep_on_device
        self.eng = net.Network.ensure_engines(1, device)[0]
        # cross-call pipelining of forward() (run_sharded): safe by default since round 4 -- the engine's internal streams wait for
        # the caller's stream, on which the windows are copied right before each call (Engine.set_pipelined); a caller whose
        # windows are resident and final before the run says so (inputs_materialised) and gets the full cross-call overlap
        self.eng.set_pipelined(bool(pipelined))
        self.input_ready = 'materialised' if inputs_materialised else None

    def _ids(self, f):
        return [min(max(f - self.t // 2 + k, 0), self.nframes - 1) for k in range(self.t)]

    def _out(self, r):
        return r if self.keep else r.cpu()

    def __call__(self, lrs, refs, first, f=None):
        ids = None if f is None else self._ids(f)
        return self._out(self.net(lrs[None].to(self.dev), refs[None].to(self.dev), first, frame_ids=ids,
                                  input_ready=self.input_ready if ids is not None else None)['result'][0])

    def phase_a(self, lrs, refs, f, hint):
        return self.net.Network.phase_a(lrs[None].to(self.dev), refs[None].to(self.dev), frame_ids=self._ids(f), first_hint=hint)

    def phase_b(self, handles, first, after_state=None):
        return self._out(self.net.Network.phase_b(handles, first, after_state=after_state)['result'][0])

    def phase_b1(self, handles, first):
        return self.net.Network.phase_b1(handles, first)

    def phase_b2(self, handles):
        return self._out(self.net.Network.phase_b2(handles)['result'][0])

    def sync(self):
        import torch as _t
        _t.cuda.synchronize(self.dev)

    def state_nbytes(self):
        return self.eng.state_nbytes(self.h, self.w)

    def export_state_packed(self):
        return self.eng.export_state_packed()

    def import_state_packed(self, buf):
        self.eng.import_state_packed(buf.to(self.dev))
