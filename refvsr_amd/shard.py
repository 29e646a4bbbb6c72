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
import os

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
    if async_op:
        # (round 6: the chain may issue a hand-off while lane a still has context messages to send -- a blocking send there can close
        #  a cycle with a peer that waits for one of them; the work handles keep their buffers alive)
        bufs = [meta] + [state[k].contiguous().to(device) for k in STATE_KEYS]
        return _Works([dist.isend(b_, dst) for b_ in bufs], bufs)
    dist.send(meta, dst)
    for k in STATE_KEYS:
        dist.send(state[k].contiguous().to(device), dst)
    return None


class _Works(object):
    """Several isend handles (and the buffers they read) behind one wait()."""

    def __init__(self, works, bufs):
        self.works, self.bufs = works, bufs

    def wait(self):
        for w_ in self.works:
            w_.wait()
        self.bufs = None


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


def _import(executor, src, channels, device, split=False):
    """Receive the state from rank src into the executor (packed single-message form when the executor offers it; split=True:
    the two-message form -- the executor starts its forward-branch step on the small message and waits for the 2x state inside it)."""
    if split:
        nb_head, nb_tail = executor.state_split_nbytes()
        head = torch.empty(nb_head, dtype=torch.uint8, device=device)
        tail = executor.state_tail_buffer(device)
        assert tail.dtype == torch.uint8 and tail.numel() == nb_tail
        w_head = dist.irecv(head, src)
        w_tail = dist.irecv(tail, src)
        w_head.wait()
        executor.import_state_split(head, tail, w_tail.wait)
        return
    nb = executor.state_nbytes() if hasattr(executor, 'state_nbytes') else None
    st = recv_state(src, channels, device, nb)
    if nb is not None:
        executor.import_state_packed(st)
    else:
        executor.import_state(st)


def _export(executor):
    return executor.export_state_packed() if hasattr(executor, 'export_state_packed') else executor.export_state()


def _send_split(executor, dst, device, keep):
    """The state as two messages, small one first (Engine.export_state_split); returns the two work handles."""
    head, tail = executor.export_state_split()
    out = []
    for buf in (head, tail):
        buf = buf if str(buf.device).startswith(str(device)) else buf.to(device)
        keep.append(buf)
        out.append(dist.isend(buf, dst))
    return out


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


def as_blocks(parts, world=None):
    """Normalise a partition to chain order: [(start, end, rank), ...] sorted by start, empty ranges dropped.  Accepts the
    per-rank form [(start, end)] * world (contiguous shards: rank = position) or a block list [(start, end, rank), ...]
    (a rank may own several blocks -- `partition_cyclic`)."""
    parts = list(parts)
    if parts and len(parts[0]) == 2:
        blocks = [(a, b, r) for r, (a, b) in enumerate(parts) if b > a]
    else:
        blocks = [(a, b, r) for a, b, r in parts if b > a]
    blocks.sort()
    for (a0, b0, _), (a1, _, _) in zip(blocks, blocks[1:]):
        assert b0 == a1, 'partition must cover the clip contiguously'
    return blocks


def partition_cyclic(nframes, world, size):
    """Block-cyclic partition: blocks of `size` consecutive frames dealt to the ranks round-robin (the last block may be
    shorter).  For a clip WITHOUT forward-branch restarts every rank then has part of the EARLY frames: the B1 chain does not
    wait for one rank to walk a whole shard at the rate of its phase A (the start-up that bounds contiguous shards), at the price
    of one cold window (two frames prepared twice) and one hand-off per block."""
    assert size >= 1
    return [(s0, min(s0 + size, nframes), (i % world)) for i, s0 in enumerate(range(0, nframes, size))]


def partition_cyclic_growing(nframes, world, t_window, t_chain_step, t_start, scale=1.0, cap=8):
    """Block-cyclic partition whose blocks GROW along the clip, for clips without forward-branch restarts.  The B1 chain reaches
    frame f about t_start + f * t_chain_step after the start; the owner of the block that starts there must have finished phase A
    of all its earlier frames (f / world of them) and of this block by then: n <= (t_start + f t_chain_step) / t_window - f / world.
    One-frame blocks get the chain going (its first step waits for ONE window, not a block of them), larger blocks later save
    hand-offs -- the chain is what bounds such a clip.  scale: safety factor on n (choose_partition tries several)."""
    import math
    assert t_window > 0 and cap >= 1
    out, f0, k = [], 0, 0
    while f0 < nframes:
        n = int(math.floor(scale * ((t_start + f0 * t_chain_step) / t_window - f0 / float(world))))
        n = max(1, min(n, cap, nframes - f0))
        out.append((f0, f0 + n, k % world))
        f0 += n
        k += 1
    return out


def window_ids(f, nframes, frame_num):
    """Frame indices of the window whose output frame is f (clip edges replicate frames, datasets.py:233-234)."""
    return [min(max(f - frame_num // 2 + k, 0), nframes - 1) for k in range(frame_num)]


def window_is_hinted(f, reset_branch):
    """Phase A of window f prepares ALL its frames (not only centre .. last): the forward branch restarts there (clip start or a
    multiple of reset_branch, RefVSR.py:168-176) and walks frames 0 .. centre itself."""
    return f == 0 or bool(reset_branch and f % reset_branch == 0)


class ContextPlan(object):
    """Who prepares which per-frame context and who needs it, for run_wavefront(exchange_contexts=True).

    A per-frame context is everything that is a function of ONE (lr_i, ref_i) pair -- matching, reference encoders, both
    aligned-attention outputs (Engine.prepare_frame, RefVSR.py:196-204,233-234,127,136): ~35 % of a frame's phase A.  Window f
    needs the contexts of its frames centre .. last (all of them when the forward branch restarts at f).  Without the exchange a
    rank prepares every context its windows need -- the first window of every block pays for two (four at a restart) contexts
    that the neighbouring block's owner prepares as well.  With it, context i is prepared ONCE, by the owner of output frame i,
    and sent to the other ranks whose windows need it (one message of ~32 MB at 270p, point to point, off the B1 chain).

    tasks[r]     lane-a order of rank r: ('prep', i) | ('a1', f): windows in block order, a context right before its first own use
                 or earlier when another rank's window needs it earlier on the B1 chain (see __init__)
    consumers[i] ranks other than the owner that need context i (sorted)
    imports[r][f] contexts rank r must have received before phase A of window f (first use only)
    Every rank issues its messages in increasing frame order (`messages`, `program`): one global order = no deadlock on in-order
    transports."""

    def __init__(self, nframes, world, parts, reset_branch, frame_num, lead=3):
        blocks = as_blocks(parts, world)
        t, ctr = frame_num, frame_num // 2
        self.blocks, self.nframes, self.world, self.t = blocks, nframes, world, t
        self.owner = {}
        for a, b, r in blocks:
            for f in range(a, b):
                self.owner[f] = r
        assert sorted(self.owner) == list(range(nframes)), 'partition must cover every frame once'
        self.windows = {r: [f for a, b, q in blocks if q == r for f in range(a, b)] for r in range(world)}
        self.needed = {}
        for f in range(nframes):
            ids = window_ids(f, nframes, t)
            self.needed[f] = sorted(set(ids[0 if window_is_hinted(f, reset_branch) else ctr:]))
        # first use of every context per rank: (window position, window)
        first_use = {r: {} for r in range(world)}
        for r in range(world):
            for pos, f in enumerate(self.windows[r]):
                for i in self.needed[f]:
                    first_use[r].setdefault(i, (pos, f))
        self.consumers = {i: sorted(r for r in range(world) if r != self.owner[i] and i in first_use[r]) for i in range(nframes)}
        self.imports = {r: {} for r in range(world)}
        for r in range(world):
            for i, (pos, f) in first_use[r].items():
                if self.owner[i] != r:
                    self.imports[r].setdefault(f, []).append(i)
            for f in self.imports[r]:
                self.imports[r][f].sort()
        # Lane-a order.  The windows of a rank keep their block order (the window cache follows the sliding window); a context is
        # prepared right before the first OWN window that needs it -- or earlier, when another rank needs it earlier.  "Earlier" is
        # measured on the B1 chain, which paces everything: window f's phase A is needed when the chain gets there, ct(f) = f - (the
        # last restart <= f) steps after its chain started (restart units are independent chains that start together); a context
        # for a foreign window f' has to be on its way `lead` steps before that (its transfer + the consumer's phase A).
        ct = {}
        for f in range(nframes):
            ct[f] = 0 if window_is_hinted(f, reset_branch) else ct[f - 1] + 1
        self.tasks = {}
        for r in range(world):
            own = [i for i in range(nframes) if self.owner[i] == r]
            seq = self.windows[r]
            due = {}
            for i in own:
                pos = first_use[r][i][0] if i in first_use[r] else len(seq)
                when = float(ct[seq[pos]]) if pos < len(seq) else float('inf')
                for q in self.consumers[i]:
                    dl = ct[first_use[q][i][1]] - lead
                    p2 = next((p for p, f in enumerate(seq) if ct[f] > dl), len(seq))
                    if p2 < pos or (p2 == pos and dl < when):
                        pos, when = min(pos, p2), min(when, float(dl))
                due[i] = (pos, when, i)
            todo = sorted(own, key=lambda i: due[i])
            out, k = [], 0
            for pos, f in enumerate(seq):
                while k < len(todo) and due[todo[k]][0] <= pos:
                    out.append(('prep', todo[k]))
                    k += 1
                out.append(('a1', f))
            assert k == len(todo), 'a context nobody needs'
            self.tasks[r] = out

    def program(self, r):
        """The host program of rank r's lane a: ('prep', i) | ('post_recv', i, peer) | ('send', i, peer) | ('wait_recv', i, peer) |
        ('a1', f), in issue order.  tasks[r] with the messages put in: a context is sent right after its preparation, a receive is
        posted (and waited for) right before the first window that needs it -- and EVERY rank issues its messages in increasing
        frame order (`messages`): whatever that order puts before a message is issued first (a receive is just posted early; a send
        whose context is not prepared yet prepares it on the spot).  One global order on all ranks = no deadlock on any in-order
        transport: per rank pair (lazily initialised ProcessGroupNCCL: a communicator and a stream per pair) and even when all
        operations of a rank share ONE stream (eagerly initialised groups serialise unbatched send / recv with everything else)."""
        msgs = self.messages(r)
        state = {'ptr': 0}
        prepared, out = set(), []

        def prepare(i):
            if i not in prepared:
                out.append(('prep', i))
                prepared.add(i)

        def issue_until(upto):
            while state['ptr'] < len(msgs) and msgs[state['ptr']][0] <= upto:
                i, kind, p = msgs[state['ptr']]
                state['ptr'] += 1
                if kind == 'recv':
                    out.append(('post_recv', i, p))
                else:
                    prepare(i)
                    out.append(('send', i, p))

        for kind, x in self.tasks[r]:
            if kind == 'prep':
                prepare(x)
                if self.consumers[x]:
                    issue_until(x)
            else:
                for i in self.imports[r].get(x, ()):
                    issue_until(i)
                    out.append(('wait_recv', i, self.owner[i]))
                out.append(('a1', x))
        issue_until(self.nframes)
        return out

    def messages(self, r):
        """All context messages of rank r in the order it issues them: [(frame, 'send' | 'recv', peer)] by frame, then peer."""
        out = []
        for i in range(self.nframes):
            if self.owner[i] == r:
                out += [(i, 'send', q) for q in self.consumers[i]]
            elif r in self.consumers[i]:
                out.append((i, 'recv', self.owner[i]))
        return out

    def pair_order(self, r, peer):
        """All context messages between r and peer, in the order both ends issue them: [(frame, 'send' | 'recv')] from r's view."""
        out = []
        for i in range(self.nframes):
            if self.owner[i] == r and peer in self.consumers[i]:
                out.append((i, 'send'))
            elif self.owner[i] == peer and r in self.consumers[i]:
                out.append((i, 'recv'))
        return out


def group_lane_ops(program, group):
    """Phase-A groups (round 6): the lane-a program of a rank -- ('a1', f) ops, with the context exchange also ('prep', i),
    ('send', i, peer), ('post_recv', i, peer), ('wait_recv', i, peer) -- with runs of up to `group` windows turned into ONE
    ('ag', (f1, ..., fk)) op: their backward branches are independent chains over identical weights and run as multi-map launches
    (Engine.phase_a_group), whether the windows are consecutive or not.  Everything that stood between the windows of a group is
    issued BEFORE the group (a window computes nothing another op waits for: deferring it reorders no message -- every rank still
    issues its sends and receives in the one global order of ContextPlan.program -- and cannot close a cycle).  group <= 1: unchanged."""
    if group <= 1:
        return list(program)
    out, cur = [], []
    for op in program:
        if op[0] == 'a1':
            cur.append(op[1])
            if len(cur) == group:
                out.append(('ag', tuple(cur)))
                cur = []
        else:
            out.append(op)
    if cur:
        out.append(('ag', tuple(cur)))
    return out


def group_time(k, group, t_a, t_a_single):
    """Phase-A time PER FRAME of a group of k windows when a full group of `group` costs t_a per frame and a lone window t_a_single:
    linear in between (the multi-map launches share one weight fill and one launch among k tiles)."""
    if group <= 1 or t_a_single is None or k >= group:
        return t_a
    return t_a_single - (t_a_single - t_a) * (k - 1) / float(group - 1)


def simulate_wavefront(nframes, world, parts, reset_branch, t_a, t_b1, t_b2, t_handoff=0.0, t_cold=0.0, interleaved=True, dt=0.01,
                       exchange=None, frame_num=5, group=1, t_a_single=None):
    """Makespan of run_wavefront from per-frame phase times (any time unit): every rank executes ONE task at a time with
    priorities  B1 (its phase A done, the previous frame's B1 done and -- across ranks -- handed over) > phase A (lane-a order)
    > B2, pre-emptively (the kernels of the two streams interleave at ~10 us granularity).
    t_cold: extra phase-A time of the first frame of a block that does not start the clip (its window is not in the cache: TWO more
    contexts to prepare; FOUR when the forward branch restarts at that frame -- a reset-aligned block start prepares the whole
    window -- i.e. 2 t_cold).
    interleaved=False: the round-3 order (B1 of a rank only after ALL its phase A).
    exchange: None | dict(t_prep, t_ctx, t_cold_x=0, lookahead=1) -- run_wavefront(exchange_contexts=True): every context is
    prepared once, by the owner of its frame (t_prep of the t_a), and sent to the ranks that need it (t_ctx after its preparation
    it is usable there); lane a follows ContextPlan.tasks (in order: a window waiting for an import blocks the lane), a block start
    costs t_cold_x (the flows of a cold window) instead of t_cold.
    group > 1 (round 6): lane a runs phase A in groups of up to `group` windows (group_lane_ops: what run_wavefront(group=) executes);
    t_a is then the per-frame phase-A time INSIDE a full group, t_a_single that of a lone window (group_time), a group is ready when
    every import of every member is there and all its windows are done at its end.  Returns the makespan."""
    blocks = as_blocks(parts)
    owner, first = {}, set()
    for a, b, r in blocks:
        first.add(a)
        for f in range(a, b):
            owner[f] = r
    assert sorted(owner) == list(range(nframes)), 'partition must cover every frame once'
    frames_of = {r: [f for f in range(nframes) if owner[f] == r] for r in range(world)}
    if exchange is None:
        cold = lambda f: 0.0 if (f not in first or f == 0) else (2.0 * t_cold if window_is_hinted(f, reset_branch) else t_cold)
        lane = {r: [('a1', f, t_a + cold(f)) for f in frames_of[r]] for r in range(world)}
        imports = {r: {} for r in range(world)}
        t_ctx = 0.0
    else:
        plan = ContextPlan(nframes, world, blocks, reset_branch, frame_num) if exchange.get('plan') is None else exchange['plan']
        t_prep, t_ctx, t_cold_x = exchange['t_prep'], exchange.get('t_ctx', 0.0), exchange.get('t_cold_x', 0.0)
        assert 0.0 <= t_prep <= t_a
        lane = {r: [(k, x, t_prep if k == 'prep' else (t_a - t_prep) + (t_cold_x if (x in first and x != 0) else 0.0))
                    for k, x in (op[:2] for op in plan.program(r) if op[0] in ('prep', 'a1'))] for r in range(world)}
        imports = plan.imports
    if group > 1:
        # the grouped lanes: ('ag', members, sum of the members' times at the group's per-frame rate); everything else unchanged
        glane = {}
        for r in range(world):
            dur = {}
            for k, x, d in lane[r]:
                if k == 'a1':
                    dur[x] = d - t_a if exchange is None else d - (t_a - exchange['t_prep'])      # the window's extras (cold terms)
            out = []
            for op in group_lane_ops([(k, x) for k, x, _ in lane[r]], group):
                if op[0] == 'ag':
                    per = group_time(len(op[1]), group, t_a, t_a_single) - (0.0 if exchange is None else exchange['t_prep'])
                    out.append(('ag', op[1], sum(per + dur[f] for f in op[1])))
                else:
                    out.append((op[0], op[1], next(d for k, x, d in lane[r] if k == op[0] and x == op[1])))
            glane[r] = out
        lane = glane
    rem_lane = {r: [d for _, _, d in lane[r]] for r in range(world)}
    rem_b1 = {f: t_b1 for f in range(nframes)}
    rem_b2 = {f: t_b2 for f in range(nframes)}
    done_a, done_b1, done_b2, done_prep = {}, {}, {}, {}
    nxt_l = {r: 0 for r in range(world)}
    nxt_b1 = {r: 0 for r in range(world)}
    nxt_b2 = {r: 0 for r in range(world)}
    n_steps = 0
    t = 0.0
    limit = 100.0 * nframes * (t_a + t_b1 + t_b2 + 2.0 * t_cold + t_handoff + t_ctx) + 1.0
    while len(done_b2) < nframes:
        assert t < limit, 'schedule does not terminate'
        for r in range(world):
            fs = frames_of[r]
            ran = False
            if nxt_b1[r] < len(fs):
                f = fs[nxt_b1[r]]
                ok = f in done_a
                if ok and needs_handoff(f, reset_branch):
                    pf = f - 1
                    ok = pf in done_b1 and t >= done_b1[pf] + (t_handoff if owner[pf] != r else 0.0)
                if ok and not interleaved:
                    ok = nxt_l[r] >= len(lane[r])
                if ok:
                    rem_b1[f] -= dt
                    if rem_b1[f] <= 1e-9:
                        done_b1[f] = t + dt
                        nxt_b1[r] += 1
                    ran = True
            if not ran and nxt_l[r] < len(lane[r]):
                kind, x, _ = lane[r][nxt_l[r]]
                ready = True
                members = (x,) if kind == 'a1' else (x if kind == 'ag' else ())
                for f_ in members:
                    for i in imports[r].get(f_, ()):
                        ready = ready and i in done_prep and t >= done_prep[i] + t_ctx
                if ready:
                    j = nxt_l[r]
                    rem_lane[r][j] -= dt
                    if rem_lane[r][j] <= 1e-9:
                        if kind == 'prep':
                            done_prep[x] = t + dt
                        for f_ in members:
                            done_a[f_] = t + dt
                        nxt_l[r] += 1
                    ran = True
            if not ran and nxt_b2[r] < nxt_b1[r]:
                f = fs[nxt_b2[r]]
                rem_b2[f] -= dt
                if rem_b2[f] <= 1e-9:
                    done_b2[f] = t + dt
                    nxt_b2[r] += 1
        n_steps += 1
        t = n_steps * dt
    return t


def predicted_speedup(nframes, world, parts, reset_branch, t_a, t_b1, t_b2, t_handoff=0.0, t_cold=0.0, interleaved=True, exchange=None,
                      frame_num=5, steps_per_frame=2000.0, group=1, t_a_single=None):
    """(speedup over one rank, makespan) of run_wavefront for a partition (per-rank ranges or a block list) from per-frame
    phase times: simulate_wavefront against nframes * (t_a + t_b1 + t_b2) -- one rank walking the clip phase by phase at the same
    per-frame times (with group > 1: in full groups).  bench.py quotes the makespan against the N = 1 HEADLINE rate as well."""
    dt = max(1e-6, (t_a + t_b1 + t_b2) / steps_per_frame)
    span = simulate_wavefront(nframes, world, parts, reset_branch, t_a, t_b1, t_b2, t_handoff, t_cold, interleaved, dt, exchange, frame_num,
                              group, t_a_single)
    seq = nframes * (t_a + t_b1 + t_b2)
    return (seq / span if span > 0 else 0.0), span


def choose_partition(nframes, world, reset_branch, t_a, t_b1, t_b2, t_handoff=0.3, t_cold=None, exchange=None, frame_num=5, group=1,
                     t_a_single=None):
    """The partition run_wavefront should use for these phase times: the best of the reset-aligned hybrid (when the forward
    branch restarts), the balanced and the growing contiguous shards and the block-cyclic partitions with 1..8 frames per
    block, by simulated makespan.  t_cold defaults to 0.85 t_a (two more frames to prepare: ~5.0 of 5.8 ms for RefVSR_small
    at 270p).  exchange: see simulate_wavefront (the context exchange of run_wavefront).  Returns (block list, predicted speedup,
    name)."""
    if t_cold is None:
        t_cold = 0.85 * t_a
    cands = []
    if reset_branch:
        cands.append(('hybrid_reset_aligned', partition_hybrid(nframes, world, reset_branch)))
    cands.append(('balanced', partition(nframes, world)))
    if nframes >= world:
        cands.append(('growing_shards', partition_chain(nframes, world, t_b1 / t_a if t_a > 0 else 0.165)))
    for size in range(1, 9):
        if size * world < nframes:
            cands.append(('block_cyclic_%d' % size, partition_cyclic(nframes, world, size)))
    if nframes > world:
        # growing blocks: phase-A time of a window and of the chain's first step with / without the context exchange
        t_w = t_a if exchange is not None else t_a + 0.5 * t_cold
        t_0 = (t_a + exchange['t_prep']) if exchange is not None else t_a
        for sc in (0.8, 1.0, 1.2):
            cands.append(('block_cyclic_growing_%.1f' % sc, partition_cyclic_growing(nframes, world, t_w, t_b1 + 0.5 * t_handoff, t_0, sc)))
    # rank the candidates at a quarter of the time resolution, re-simulate the best three at the full one
    rough = []
    for name, parts in cands:
        sp, _ = predicted_speedup(nframes, world, parts, reset_branch, t_a, t_b1, t_b2, t_handoff, t_cold, True, exchange, frame_num, 500.0,
                                  group, t_a_single)
        rough.append((sp, name, parts))
    rough.sort(key=lambda v: -v[0])
    fine = []
    for _, name, parts in rough[:3]:
        sp, _ = predicted_speedup(nframes, world, parts, reset_branch, t_a, t_b1, t_b2, t_handoff, t_cold, True, exchange, frame_num,
                                  group=group, t_a_single=t_a_single)
        fine.append((as_blocks(parts), sp, name))
    top = max(v[1] for v in fine)
    # within 1 % of the best: the partition with the fewest blocks (fewest hand-offs and messages: the model's terms for those are
    # the assumed ones)
    return min((v for v in fine if v[1] >= 0.99 * top), key=lambda v: (len(v[0]), -v[1]))


_CTX_GROUP = None


def context_group():
    """A second process group for the context messages of run_wavefront(exchange_contexts=True): its own RCCL communicators and
    streams, so that a hand-off of the B1 chain never queues behind a 32 MB context transfer between the same two ranks.
    Collective on first use (every rank calls run_wavefront)."""
    global _CTX_GROUP
    if _CTX_GROUP is None:
        _CTX_GROUP = dist.new_group(ranks=list(range(dist.get_world_size())))
    return _CTX_GROUP


class _NullCtx(object):
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def run_wavefront(executor, get_window, nframes, frame_num, reset_branch, channels, device, on_result=None, parts=None,
                  timings=None, exchange_contexts=False, group=1):
    """See _run_wavefront (this wrapper only makes sure the executor's strict-context mode ends with the run)."""
    try:
        return _run_wavefront(executor, get_window, nframes, frame_num, reset_branch, channels, device, on_result, parts, timings,
                              exchange_contexts, group)
    finally:
        if exchange_contexts and hasattr(executor, 'strict_contexts'):
            executor.strict_contexts(False)


def _run_wavefront(executor, get_window, nframes, frame_num, reset_branch, channels, device, on_result=None, parts=None,
                   timings=None, exchange_contexts=False, group=1):
    """Two-phase run of this rank's share of the clip.  parts: per-rank ranges [(start, end)] * world or a block list
    [(start, end, rank), ...] (default: the balanced contiguous partition).

    executor.phase_a(lrs, refs, frame_index, first_hint) -> handle   (state-free: preparation + backward branch)
    executor.phase_b1(handle, is_first_frame) -> handle              (forward-branch step: carries the state; in frame order)
    executor.phase_b2(handle) -> result                              (BW/FW fusion + upsampler: state-free)

    Order of issue on every rank (round 4): phase A of ALL local frames | per block in chain order: receive the state, B1 of the
    block's frames, send the state | B2 of all local frames.  An executor with two execution lanes (EngineExecutor: two HIP
    streams, `lane_a()` / `lane_b()` + the mark / wait hooks) runs the B1 chain on the second lane: B1(f) starts as soon as
    ITS phase A is done and the state is there, while the first lane is still walking the phase A of later frames -- the serial
    chain over the ranks no longer waits for a rank's whole phase A (round 3: `interleaved=False` of the model).  Executors
    without lanes (the CPU oracle on gloo) run the three groups one after the other: same results, same messages.
    Executors without phase_b1 / phase_b2 run phase_b(handle, first) per frame (B2 then sits on the chain).
    exchange_contexts (executors with prepare_context / export_context / import_context): every per-frame context is prepared
    ONCE, by the owner of its output frame, and sent to the other ranks whose windows need it (ContextPlan) -- the first window
    of a block no longer prepares two (four at a restart) contexts its neighbour prepares as well.  Lane a then walks
    ContextPlan.program(rank): prepare context i (+ isend to its consumers) | receive what window f lacks, then phase A of f (which
    finds every context prepared: a missing one raises, executor.strict_contexts).  Context messages travel in their own process
    group; every rank issues them in increasing frame order.
    group > 1 (round 6; executors with phase_a_group(windows, frames, hints) -> handles): lane a runs phase A in groups of up to
    `group` local windows (group_lane_ops) -- their backward branches as multi-map launches; B1(f) then waits for the group that
    holds f.  Same results, same messages in the same order.
    Results are identical to the sequential run.  Returns {frame: result} for the local frames.  timings (optional dict):
    'issue_a' / 'recv_wait' / 'issue_b1' / 'issue_b2' host seconds, 'handoff_messages' sent, 'blocks' local blocks,
    'context_messages' sent, 'context_wait' host seconds blocked waiting for contexts."""
    import time
    # (no process group: ONE rank walking the clip through the executor -- bench.py's one-rank figure; nothing is sent or received)
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    blocks = as_blocks(parts if parts is not None else partition(nframes, world), world)
    mine = [(a, b) for a, b, r in blocks if r == rank]
    owner_of = {}
    for a, b, r in blocks:
        for f in range(a, b):
            owner_of[f] = r
    if hasattr(executor, 'begin_run'):
        executor.begin_run()                                       # (two-lane executors: the lanes wait for the caller's stream)
    lane_a = getattr(executor, 'lane_a', _NullCtx)
    lane_b = getattr(executor, 'lane_b', _NullCtx)
    lane_c = getattr(executor, 'lane_c', lane_a)                   # the upsamplers' lane (EngineExecutor: M, behind the backward chains)
    mark = getattr(executor, 'mark', lambda what, f: None)          # record "what of frame f is enqueued up to here"
    wait = getattr(executor, 'wait', lambda what, f: None)          # the CURRENT lane waits for that mark
    tim = {'issue_a': 0.0, 'recv_wait': 0.0, 'issue_b1': 0.0, 'issue_b2': 0.0, 'handoff_messages': 0, 'blocks': len(mine),
           'context_messages': 0, 'context_wait': 0.0}
    t0 = time.perf_counter()
    handles, keep = {}, []
    results, pending = {}, []
    split = hasattr(executor, 'phase_b1')
    # the hand-off in two messages (the receiver's step starts on the small one): executors that offer it, with the B1 / B2 split
    two_msg = split and bool(getattr(executor, 'split_handoff', False))
    # ---- the chain (B1 / B2 executors).  Round 6: the host no longer issues it after ALL of lane a -- after every window (group) of
    # lane a, advance(False) issues the forward-branch steps that can go WITHOUT a blocking receive (a block that starts at a restart
    # or continues a local block: every block of a reset-aligned partition) and, one call later, their upsamplers, so that on shards
    # of tens of frames lane b and the upsamplers run under lane a instead of behind it (a 32-frame shard: the host needs > 60 ms
    # to issue its phase A).  Blocks in frame order, as before (the engine carries ONE state); a block that needs a remote state
    # stops the early issue until lane a is through (a blocking receive in the middle of lane a could close a cycle with the context
    # messages this rank has not sent yet).  Sends are asynchronous wherever they are issued.
    chain = [f for a, b in mine for f in range(a, b)]
    block_of = {f: (a, b) for a, b in mine for f in range(a, b)}
    cst = {'b1': 0, 'b2': 0, 'first': False}
    early_chain = split and bool(getattr(executor, 'early_chain', hasattr(executor, 'lane_b')))

    def advance(blocking):
        b2_upto = len(chain) if blocking else cst['b1']            # upsamplers lag one call: their B1 is long done when lane a gets there
        t2 = time.perf_counter()
        while cst['b1'] < len(chain):
            f = chain[cst['b1']]
            a, b = block_of[f]
            if f not in handles:
                break                                              # its phase A is not issued yet
            with lane_b():
                if f == a:
                    if needs_handoff(a, reset_branch):
                        if owner_of[a - 1] != rank:
                            if not blocking:
                                break
                            t1 = time.perf_counter()
                            _import(executor, owner_of[a - 1], channels, device, split=two_msg)
                            tim['recv_wait'] += time.perf_counter() - t1
                        cst['first'] = False
                    else:
                        cst['first'] = True
                wait('a', f)
                handles[f] = executor.phase_b1(handles[f], cst['first'])
                mark('b1', f)
                cst['first'] = False
                nxt_rank = owner_of.get(b)                         # owner of the block that follows in chain order
                if f == b - 1 and nxt_rank is not None and nxt_rank != rank and needs_handoff(b, reset_branch):
                    if two_msg:                                    # the next block's chain starts as soon as this one ends
                        pending.extend(_send_split(executor, nxt_rank, device, keep))
                    else:
                        pending.append(send_state(_export(executor), nxt_rank, device, async_op=True))
                    tim['handoff_messages'] += 1
            cst['b1'] += 1
        t3 = time.perf_counter()
        tim['issue_b1'] += t3 - t2
        if blocking:
            b2_upto = cst['b1']
        with lane_c():
            while cst['b2'] < min(b2_upto, cst['b1']):             # ---- B2: upsamplers, off the chain
                f = chain[cst['b2']]
                wait('b1', f)
                h = handles.pop(f)
                keep.append(h)                                     # (two lanes: the handle's tensors stay allocated until the
                out = executor.phase_b2(h)                         #  final synchronisation -- they are read on both streams)
                results[f] = out
                if on_result is not None:
                    on_result(f, out)
                cst['b2'] += 1
        tim['issue_b2'] += time.perf_counter() - t3

    group = int(group) if hasattr(executor, 'phase_a_group') else 1

    def run_group(fs, window_of):                                  # one ('ag', frames) op of the grouped lane-a program
        hs = executor.phase_a_group([window_of(f) for f in fs], list(fs), [window_is_hinted(f, reset_branch) for f in fs])
        for f, h in zip(fs, hs):
            handles[f] = h
            mark('a', f)
        if early_chain:
            advance(False)
    if not exchange_contexts:
        with lane_a():
            for op in group_lane_ops([('a1', f) for a, b in mine for f in range(a, b)], group):   # ---- phase A: no communication, no state
                if op[0] == 'ag':
                    run_group(op[1], get_window)
                else:
                    lrs, refs = get_window(op[1])
                    handles[op[1]] = executor.phase_a(lrs, refs, op[1], window_is_hinted(op[1], reset_branch))
                    mark('a', op[1])
                    if early_chain:
                        advance(False)
    else:
        plan = ContextPlan(nframes, world, blocks, reset_branch, frame_num)
        grp = context_group()
        comm_lane = getattr(executor, 'comm_lane', _NullCtx)       # an idle lane to post receives from (they then wait for nothing)
        wins, frame_in = {}, {}
        for a, b in mine:
            for f in range(a, b):
                wins[f] = get_window(f)
                for j, i in enumerate(window_ids(f, nframes, frame_num)):
                    frame_in.setdefault(i, (wins[f][0][j], wins[f][1][j]))
        rx = {}
        strict = getattr(executor, 'strict_contexts', lambda on: None)
        strict(True)                                               # a context that is missing is a bug, not something to recompute
        with lane_a():
            for op in group_lane_ops(plan.program(rank), group):
                kind, x = op[0], op[1]
                if kind == 'ag':
                    run_group(x, lambda f: wins[f])
                elif kind == 'prep':
                    executor.prepare_context(x, frame_in[x][0], frame_in[x][1])
                elif kind == 'post_recv':
                    with comm_lane():
                        buf = torch.empty(executor.context_nbytes(), dtype=torch.uint8, device=device)
                        rx[x] = (dist.irecv(buf, op[2], group=grp), buf)
                elif kind == 'send':
                    buf = executor.export_context(x)
                    buf = buf if str(buf.device).startswith(str(device)) else buf.to(device)
                    pending.append(dist.isend(buf, op[2], group=grp))
                    keep.append(buf)
                    tim['context_messages'] += 1
                elif kind == 'wait_recv':
                    wk, buf = rx.pop(x)
                    t1 = time.perf_counter()
                    wk.wait()
                    tim['context_wait'] += time.perf_counter() - t1
                    executor.import_context(x, frame_in[x][0], frame_in[x][1], buf)
                else:
                    lrs, refs = wins[x]
                    handles[x] = executor.phase_a(lrs, refs, x, window_is_hinted(x, reset_branch))
                    mark('a', x)
                    if early_chain:
                        advance(False)
        assert not rx
    tim['issue_a'] = time.perf_counter() - t0 - tim['issue_b1'] - tim['issue_b2']
    if split:
        advance(True)                                              # ---- the rest of the chain (blocking receives), then every B2 left
    else:
        for a, b in mine:                                          # ---- the chain: one block at a time, in frame order
            nxt_rank = owner_of.get(b)                             # owner of the block that follows in chain order
            handoff_out = nxt_rank is not None and nxt_rank != rank and needs_handoff(b, reset_branch)
            with lane_b():
                t1 = time.perf_counter()
                if needs_handoff(a, reset_branch):
                    if owner_of[a - 1] != rank:
                        _import(executor, owner_of[a - 1], channels, device, split=False)
                    first = False
                else:
                    first = True
                tim['recv_wait'] += time.perf_counter() - t1
                t2 = time.perf_counter()

                def start_send():    # called by phase_b of the block's LAST frame as soon as its carried state is final
                    pending.append(send_state(_export(executor), nxt_rank, device, async_op=True))
                early = handoff_out and getattr(executor, 'supports_after_state', False)
                for f in range(a, b):
                    wait('a', f)
                    h = handles.pop(f)
                    keep.append(h)
                    if early and f == b - 1:
                        out = executor.phase_b(h, first, after_state=start_send)
                    else:
                        out = executor.phase_b(h, first)
                    first = False
                    results[f] = out
                    if on_result is not None:
                        on_result(f, out)
                if handoff_out:
                    if not early:
                        send_state(_export(executor), nxt_rank, device)
                    tim['handoff_messages'] += 1
                tim['issue_b1'] += time.perf_counter() - t2
    for wk in pending:
        if wk is not None:
            wk.wait()
    getattr(executor, 'sync', lambda: None)()
    del keep[:]
    if timings is not None:
        timings.update(tim)
    return results


class EngineExecutor(object):
    """Adapter of the HIP `SRNet` (refvsr_amd/model.py) to run_sharded / run_wavefront: windows are named by their frame
    indices (id-keyed window cache, no content compare), the hand-off uses the packed single-message state."""
    supports_after_state = True

    def __init__(self, net, device, h, w, nframes, frame_num, keep_on_device=True, pipelined=True, inputs_materialised=False,
                 split_handoff=True, split_phase_a=True):
        self.net, self.dev, self.h, self.w, self.nframes, self.t = net, device, h, w, nframes, frame_num
        self.keep = keep_on_device
        self.eng = net.Network.ensure_engines(1, device)[0]
        # run_wavefront: the hand-off as [header | feat | flow | conf] + [feat_up]; the forward-branch step of the receiver starts on
        # the first message (Engine.export_state_split).  RefVSR_IR keeps the one-message form.
        self.split_handoff = bool(split_handoff) and bool(getattr(self.eng, 'split_state_ok', False))
        # cross-call pipelining of forward() (run_sharded): safe by default since round 4 -- the engine's internal streams wait for
        # the caller's stream, on which the windows are copied right before each call (Engine.set_pipelined); a caller whose
        # windows are resident and final before the run says so (inputs_materialised) and gets the full cross-call overlap
        self.eng.set_pipelined(bool(pipelined))
        self.input_ready = 'materialised' if inputs_materialised else None
        # run_wavefront's two execution lanes: phase A (+ B2) on one HIP stream, the B1 chain (+ the hand-off) on another, ordered by
        # per-frame events -- B1(f) runs as soon as phase A of ITS frame is done and the state has arrived
        self._lanes = None
        self._marks = {}
        self._marks_preset = set()
        self._ctx_spec = None
        # round 6: phase-A groups with the preparation on lane a and the backward chains (+ the upsamplers) on a third lane, like a
        # frame group's P | M sections (REFVSR_SHARD_SPLIT_A=0: the whole phase A on lane a, the A/B)
        self.split_phase_a = bool(split_phase_a) and os.environ.get('REFVSR_SHARD_SPLIT_A', '1') != '0' and self.eng.group_ok()

    def _streams(self):
        """(lane a = P: per-frame preparation, flows, context messages | lane b = F: the B1 chain + hand-off | comm | lane c = M: the
        backward chains of the phase-A groups + the upsamplers) -- the stream layout of a frame group (P | F | M)."""
        if self._lanes is None:
            self._lanes = tuple(torch.cuda.Stream(device=self.dev) for _ in range(4))
            self.begin_run()
        return self._lanes

    def begin_run(self):
        """Start of every run_wavefront on this executor: the lanes wait for the caller's stream as it stands -- whatever produced
        the (device-resident) windows / the weights there, also for the SECOND run on one executor (sync() orders the
        caller after the lanes, not the lanes after the caller)."""
        if self._lanes is not None:
            cur = torch.cuda.current_stream(self.dev)
            for s_ in self._lanes:
                s_.wait_stream(cur)

    def comm_lane(self):
        """An otherwise idle stream to post context receives from: a receive posted from lane a would first wait for everything
        lane a has queued (ProcessGroupNCCL orders its stream behind the caller's), i.e. the transfer would not overlap."""
        return torch.cuda.stream(self._streams()[2])

    # ---- per-frame contexts (run_wavefront with exchange_contexts).  Network.forward / phase_a name the frames of batch element b
    #      (b, frame id) towards its engine: the engine-level id of frame f of this single-clip executor is (0, f)
    def prepare_context(self, f, lr, ref):
        self.eng.prepare_context(lr.to(self.dev), ref.to(self.dev), (0, f))
        if self._ctx_spec is None:
            self._ctx_spec = self.eng.context_spec((0, f))

    def context_nbytes(self):
        assert self._ctx_spec is not None, 'no context prepared yet: the message layout comes from the first own context'
        return self.eng.context_nbytes(self._ctx_spec)

    def export_context(self, f):
        return self.eng.export_context((0, f))

    def strict_contexts(self, on):
        """While on, a window that finds one of its contexts missing raises instead of preparing it (Engine.prepare_frame)."""
        self.eng.ctx_strict = bool(on)

    def import_context(self, f, lr, ref, buf):
        b = buf.to(self.dev)
        if b.is_cuda:
            b.record_stream(torch.cuda.current_stream(self.dev))
        self.eng.import_context(lr.to(self.dev), ref.to(self.dev), (0, f), b, self._ctx_spec)

    def lane_a(self):
        return torch.cuda.stream(self._streams()[0])

    def lane_b(self):
        return torch.cuda.stream(self._streams()[1])

    def lane_c(self):
        """Where the upsamplers (B2) go: behind the backward chains on M when the phase-A groups are split over P | M, else lane a."""
        return torch.cuda.stream(self._streams()[3 if self.split_phase_a else 0])

    def mark(self, what, f):
        if what == 'a' and (what, f) in self._marks_preset:       # a split phase-A group: the handle's own event on M is the mark
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        self._marks[(what, f)] = ev

    def wait(self, what, f):
        ev = self._marks.get((what, f))
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)

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

    def phase_a_group(self, windows, fs, hints):
        """phase_a of several local windows in one pass (run_wavefront(group=)): windows [(lrs, refs)], fs their frame indices."""
        streams = None
        if self.split_phase_a:
            lanes = self._streams()
            streams = (lanes[3], (lanes[1],))
        hs = self.net.Network.phase_a_group([w_[0].to(self.dev) for w_ in windows], [w_[1].to(self.dev) for w_ in windows],
                                            [self._ids(f) for f in fs], hints, streams)
        for f, h in zip(fs, hs):
            if h[0].get('ready') is not None:
                self._marks[('a', f)] = h[0]['ready']
                self._marks_preset.add(('a', f))
        return hs

    def phase_b(self, handles, first, after_state=None):
        return self._out(self.net.Network.phase_b(handles, first, after_state=after_state)['result'][0])

    def phase_b1(self, handles, first):
        return self.net.Network.phase_b1(handles, first)

    def phase_b2(self, handles):
        return self._out(self.net.Network.phase_b2(handles)['result'][0])

    def sync(self):
        torch.cuda.synchronize(self.dev)
        if self._lanes is not None:                         # results were produced on the lanes: order them before the caller's stream
            cur = torch.cuda.current_stream(self.dev)
            for s_ in self._lanes:
                cur.wait_stream(s_)
        self._marks.clear()
        self._marks_preset.clear()

    def state_nbytes(self):
        return self.eng.state_nbytes(self.h, self.w)

    def export_state_packed(self):
        return self.eng.export_state_packed()

    def import_state_packed(self, buf):
        self.eng.import_state_packed(buf.to(self.dev))

    # ---- the two-message hand-off
    def state_split_nbytes(self):
        cs = self.eng._state_cs()
        return self.eng.state_head_nbytes(self.h, self.w), 2 * self.h * 2 * self.w * cs * 2

    def export_state_split(self):
        head, feat_up = self.eng.export_state_split()
        return head, feat_up.reshape(-1).view(torch.uint8)

    def state_tail_buffer(self, device):
        """The buffer the second message is received into: the engine's next 2x state itself when the transport writes device
        memory (RCCL), a host buffer otherwise (gloo: copied to the device by the deferred wait)."""
        cs = self.eng._state_cs()
        self._tail = torch.empty((2 * self.h, 2 * self.w, cs), dtype=torch.float16, device=self.dev)
        if str(device).startswith('cuda'):
            return self._tail.view(-1).view(torch.uint8)
        return torch.empty(self._tail.numel() * 2, dtype=torch.uint8, device=device)

    def import_state_split(self, head, tail, wait_tail):
        feat_up = self._tail
        if tail.is_cuda:
            waiter = wait_tail
        else:
            def waiter():
                wait_tail()
                feat_up.copy_(tail.view(torch.float16).view(feat_up.shape))
        self.eng.import_state_head(head.to(self.dev), feat_up, waiter)
