"""N>1 path on CPU: two processes, gloo backend, the oracle as executor.  The sharded run (with the
point-to-point state hand-off, and with the exchange-free reset-aligned partition) must reproduce
the sequential run bit-for-bit."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Exec(object):
    def __init__(self, cfg, sd):
        from oracle import refvsr_oracle as orc
        self.o = orc.OracleNetwork(cfg, sd)

    def __call__(self, lrs, refs, first):
        return self.o.forward(lrs[None], refs[None], first)['result'][0]

    def phase_a(self, lrs, refs, f, hint):
        return self.o.phase_a(lrs[None], refs[None], first_hint=hint)

    def phase_b(self, handle, first):
        return self.o.phase_b(handle, first)['result'][0]

    def phase_b1(self, handle, first):
        return self.o.phase_b1(handle, first)

    def phase_b2(self, handle):
        return self.o.phase_b2(handle)['result'][0]

    def export_state(self):
        return self.o.export_state()

    def import_state(self, st):
        self.o.import_state(st)


class _ExecNoSplit(_Exec):
    """An executor without the B1 / B2 split: run_wavefront falls back to phase_b per frame."""
    phase_b1 = property(lambda self: (_ for _ in ()).throw(AttributeError('phase_b1')))
    phase_b2 = property(lambda self: (_ for _ in ()).throw(AttributeError('phase_b2')))


class _ExecLanes(_Exec):
    """An executor that offers run_wavefront's two-lane hooks (EngineExecutor: two HIP streams + events) and records what ran
    where: phase A and B2 must be issued on lane a, the B1 chain and the hand-off on lane b, every B1(f) behind wait('a', f) and
    every B2(f) behind wait('b1', f)."""

    def __init__(self, cfg, sd):
        _Exec.__init__(self, cfg, sd)
        self.lane, self.log, self.marks = None, [], set()

    class _Lane(object):
        def __init__(self, ex, name):
            self.ex, self.name = ex, name

        def __enter__(self):
            self.prev, self.ex.lane = self.ex.lane, self.name

        def __exit__(self, *exc):
            self.ex.lane = self.prev
            return False

    def lane_a(self):
        return self._Lane(self, 'a')

    def lane_b(self):
        return self._Lane(self, 'b')

    def mark(self, what, f):
        self.marks.add((what, f))

    def wait(self, what, f):
        assert (what, f) in self.marks, 'wait for a mark that was never recorded: %s %d' % (what, f)
        self.log.append(('wait', what, f, self.lane))

    def phase_a(self, lrs, refs, f, hint):
        assert self.lane == 'a'
        h = _Exec.phase_a(self, lrs, refs, f, hint)
        h['_frame'] = f
        return h

    def phase_a_group(self, windows, fs, hints):
        """run_wavefront(group=): several windows per lane-a op (the engine runs their backward branches as multi-map launches; the
        oracle walks them one by one -- the protocol around them is what is under test)."""
        self.groups = getattr(self, 'groups', []) + [tuple(fs)]
        return [self.phase_a(w_[0], w_[1], f, h) for w_, f, h in zip(windows, fs, hints)]

    def phase_b1(self, handle, first):
        assert self.lane == 'b' and self.log and self.log[-1][:3] == ('wait', 'a', handle['_frame'])
        return _Exec.phase_b1(self, handle, first)

    def phase_b2(self, handle):
        assert self.lane == 'a' and self.log[-1][:3] == ('wait', 'b1', handle['_frame'])
        return _Exec.phase_b2(self, handle)

    def import_state(self, st):
        assert self.lane == 'b'
        _Exec.import_state(self, st)


class _ExecCtx(_ExecLanes):
    """+ the context hooks of run_wavefront(exchange_contexts=True).  A context of the oracle = the matching of one frame pair
    (conf map + index map: its per-frame work that windows share); phase A must find the matching of EVERY frame it needs already
    there -- prepared here or received -- and every frame's matching is computed exactly once across the ranks."""

    def __init__(self, cfg, sd, nframes, t):
        _ExecLanes.__init__(self, cfg, sd)
        self.nframes, self.t, self.ctx, self.prepared_here, self.imported_here, self.shapes = nframes, t, {}, [], [], None

    def prepare_context(self, f, lr, ref):
        assert self.lane == 'a' and f not in self.ctx
        conf, idx = self.o._feature_match(lr[None], ref[None])
        self.ctx[f] = (conf, idx)
        self.prepared_here.append(f)
        if self.shapes is None:
            self.shapes = (tuple(conf.shape), conf.dtype, tuple(idx.shape), idx.dtype)

    def context_nbytes(self):
        cs, cd, is_, id_ = self.shapes
        n = lambda shp, dt: int(torch.tensor(shp).prod()) * torch.empty((), dtype=dt).element_size()
        return n(cs, cd) + n(is_, id_)

    def export_context(self, f):
        conf, idx = self.ctx[f]
        return torch.cat([conf.contiguous().reshape(-1).view(torch.uint8), idx.contiguous().reshape(-1).view(torch.uint8)])

    def import_context(self, f, lr, ref, buf):
        assert self.lane == 'a' and f not in self.ctx and buf.numel() == self.context_nbytes()
        cs, cd, is_, id_ = self.shapes
        nc = int(torch.tensor(cs).prod()) * torch.empty((), dtype=cd).element_size()
        self.ctx[f] = (buf[:nc].clone().view(cd).view(cs), buf[nc:].clone().view(id_).view(is_))
        self.imported_here.append(f)

    def phase_a(self, lrs, refs, f, hint):
        from refvsr_amd import shard
        assert self.lane == 'a'
        ids = shard.window_ids(f, self.nframes, self.t)
        need = range(0 if hint else self.t // 2, self.t)
        assert all(ids[j] in self.ctx for j in need), 'window %d: a context is missing (%s)' % (f, [ids[j] for j in need if ids[j] not in self.ctx])
        h = self.o.phase_a(lrs[None], refs[None], first_hint=hint, contexts={j: self.ctx[ids[j]] for j in need})
        h['_frame'] = f
        return h


class _ExecSplit(_ExecLanes):
    """+ the two-message hand-off of run_wavefront (EngineExecutor.split_handoff): [iteration counter | feat | flow | conf] first,
    the 2x state second; the receiver may start its forward-branch step on the first message and must wait for the second one
    before it reads the 2x state -- here: the wait is deferred into phase_b1 and recorded."""
    split_handoff = True

    def __init__(self, cfg, sd):
        _ExecLanes.__init__(self, cfg, sd)
        self._pending, self.deferred_waits, self.shapes = None, 0, None

    def _shapes(self):
        if self.shapes is None:
            st = self.o.export_state()
            self.shapes = {k: tuple(st[k].shape) for k in ('feat', 'flow', 'conf', 'feat_up')}
        return self.shapes

    def state_split_nbytes(self):
        sh = self._probe_shapes
        n = lambda k: 4 * int(torch.tensor(sh[k]).prod())
        return 4 + n('feat') + n('flow') + n('conf'), n('feat_up')

    def export_state_split(self):
        st = self.o.export_state()
        head = torch.cat([torch.tensor([float(st['frame_itr_num'])])] + [st[k].float().reshape(-1) for k in ('feat', 'flow', 'conf')])
        return head.contiguous().view(torch.uint8), st['feat_up'].float().contiguous().reshape(-1).view(torch.uint8)

    def state_tail_buffer(self, device):
        return torch.empty(self.state_split_nbytes()[1], dtype=torch.uint8)

    def import_state_split(self, head, tail, wait_tail):
        assert self.lane == 'b'
        sh = self._probe_shapes
        v = head.view(torch.float32)
        st, o = {'frame_itr_num': int(v[0])}, 1
        for k in ('feat', 'flow', 'conf'):
            n = int(torch.tensor(sh[k]).prod())
            st[k] = v[o:o + n].clone().view(sh[k])
            o += n
        self._pending = (st, tail, wait_tail)

    def phase_b1(self, handle, first):
        if self._pending is not None:                      # the first step behind a hand-off: the 2x state is awaited HERE
            st, tail, wait_tail = self._pending
            self._pending = None
            wait_tail()
            self.deferred_waits += 1
            st['feat_up'] = tail.view(torch.float32).clone().view(self._probe_shapes['feat_up'])
            _Exec.import_state(self, st)
        return _ExecLanes.phase_b1(self, handle, first)


def _setup(reset, nframes=6, name='config_RefVSR_small_L1'):
    from refvsr_amd import get_config, make_state_dict
    from refvsr_amd.synth import make_clip, window_indices
    cfg = get_config('p', 'm', name)
    cfg.frame_num = 3
    if reset != 'keep':
        cfg.reset_branch = reset
    sd = make_state_dict(cfg, 1234, variant='plausible')
    lr, rf, _ = make_clip(nframes, 16, 16, seed=3)
    get = lambda f: (lr[window_indices(f, nframes, 3)], rf[window_indices(f, nframes, 3)])
    return cfg, sd, get


def _worker(rank, world, port, reset, aligned, q, wavefront=False, nframes=6, name='config_RefVSR_small_L1'):
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from refvsr_amd import shard
    cfg, sd, get = _setup(reset, nframes, name)
    reset = cfg.reset_branch
    info = {}
    if wavefront == 'hybrid':
        parts = shard.partition_hybrid(nframes, world, reset)
        res = shard.run_wavefront(_Exec(cfg, sd), get, nframes, 3, reset, cfg.mid_channels, 'cpu', parts=parts)
    elif wavefront == 'cyclic':                            # restart-free clip, block-cyclic partition: several blocks per rank
        parts = shard.partition_cyclic(nframes, world, 2)
        res = shard.run_wavefront(_ExecLanes(cfg, sd), get, nframes, 3, reset, cfg.mid_channels, 'cpu', parts=parts)
    elif wavefront == 'growing':                           # restart-free clip, growing contiguous shards
        parts = shard.partition_chain(nframes, world)
        tim = {}
        res = shard.run_wavefront(_ExecLanes(cfg, sd), get, nframes, 3, reset, cfg.mid_channels, 'cpu', parts=parts, timings=tim)
        assert tim['blocks'] == 1 and tim['handoff_messages'] == (1 if rank + 1 < world else 0)
    elif isinstance(wavefront, str) and wavefront.startswith('exchange'):   # per-frame contexts prepared once and exchanged
        G = 1
        if '@g' in wavefront:                              # '...@gN': lane a in phase-A groups of up to N windows (round 6)
            wavefront, g_ = wavefront.split('@g')
            G = int(g_)
        fam = wavefront.split('_', 1)[1]
        parts = {'cyclic': lambda: shard.partition_cyclic(nframes, world, 2), 'hybrid': lambda: shard.partition_hybrid(nframes, world, reset),
                 'balanced': lambda: shard.partition(nframes, world),
                 'growing': lambda: shard.partition_cyclic_growing(nframes, world, 5.3, 1.0, 7.2)}[fam]()
        ex = _ExecCtx(cfg, sd, nframes, 3)
        tim = {}
        res = shard.run_wavefront(ex, get, nframes, 3, reset, cfg.mid_channels, 'cpu', parts=parts, timings=tim, exchange_contexts=True, group=G)
        info = {'prepared': ex.prepared_here, 'imported': ex.imported_here, 'messages': tim['context_messages'], 'groups': getattr(ex, 'groups', [])}
    elif wavefront == 'two_message':                       # the hand-off as two messages, the second awaited inside the step
        ex = _ExecSplit(cfg, sd)
        C, hh = cfg.mid_channels, 16
        ex._probe_shapes = {'feat': (C, hh, hh), 'flow': (2, hh, hh), 'conf': (1, hh, hh), 'feat_up': (C, 2 * hh, 2 * hh)}    # (batch dim dropped)
        tim = {}
        res = shard.run_wavefront(ex, get, nframes, 3, reset, cfg.mid_channels, 'cpu', parts=shard.partition_cyclic(nframes, world, 2), timings=tim)
        nblocks = len([1 for a, b, r in shard.partition_cyclic(nframes, world, 2) if r == rank])
        assert ex.deferred_waits == nblocks - (1 if rank == 0 else 0), (ex.deferred_waits, nblocks)
    elif wavefront == 'nosplit':
        res = shard.run_wavefront(_ExecNoSplit(cfg, sd), get, nframes, 3, reset, cfg.mid_channels, 'cpu')
    elif wavefront:
        res = shard.run_wavefront(_Exec(cfg, sd), get, nframes, 3, reset, cfg.mid_channels, 'cpu')
    else:
        res = shard.run_sharded(_Exec(cfg, sd), get, nframes, 3, reset, cfg.mid_channels, 'cpu', aligned=aligned)
    q.put((rank, {f: v.clone().numpy() for f, v in res.items()}, info))     # by value: no fd passing after exit
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(reset, aligned, wavefront=False, world=2, nframes=6, name='config_RefVSR_small_L1'):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, reset, aligned, q, wavefront, nframes, name)) for r in range(world)]
    for p in procs:
        p.start()
    got, infos = {}, {}
    try:
        for _ in range(world):
            rank, res, info = q.get(timeout=300)
            got.update({f: torch.from_numpy(v) for f, v in res.items()})
            infos[rank] = info
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:
        for p in procs:                      # a worker that died leaves its peers waiting in a receive: never leave them behind
            if p.is_alive():
                p.kill()
                p.join(10)
    cfg, sd, get = _setup(reset, nframes, name)
    ex = _Exec(cfg, sd)
    nthr = torch.get_num_threads()
    torch.set_num_threads(2)                 # same CPU kernel blocking as the workers => bit-exact compare
    try:
        for f in range(nframes):
            want = ex(*get(f), f == 0)
            assert torch.equal(got[f], want), 'frame %d differs from the sequential run' % f
    finally:
        torch.set_num_threads(nthr)
    if isinstance(wavefront, str) and wavefront.startswith('exchange'):
        return got, infos
    return got


def test_handoff_partition_matches_sequential():
    _run(reset=None, aligned=False)          # boundary at frame 3: needs the state hand-off


def test_reset_aligned_partition_is_exchange_free():
    _run(reset=3, aligned=True)              # boundary at a multiple of reset_branch: no communication


def test_wavefront_matches_sequential():
    """phase A of all frames first, phase B behind the hand-off: boundary at frame 3 (no reset), bit-identical."""
    _run(reset=None, aligned=False, wavefront=True)


def test_wavefront_with_reset_inside_a_shard():
    """reset_branch=4: frame 4 (inside rank 1's shard, which starts behind a hand-off) restarts the forward branch."""
    _run(reset=4, aligned=False, wavefront=True)


def test_world8_64_frame_clip_reset9_wavefront():
    """BASELINE configs[3] in miniature: config_RefVSR_small_MFID (reset_branch = 9), a 64-frame clip sharded over 8
    ranks (8 frames each: every shard contains a restart, all but the first start behind a hand-off), wavefront schedule
    over gloo -- bit-identical to the sequential stream of all 64 frames."""
    got = _run(reset='keep', aligned=False, wavefront=True, world=8, nframes=64, name='config_RefVSR_small_MFID')
    assert sorted(got) == list(range(64))


def test_wavefront_without_split_executor():
    """Executors that only offer phase_b keep working (B2 stays on the chain)."""
    _run(reset=None, aligned=False, wavefront='nosplit')


def test_world8_64_frame_clip_reset9_hybrid_partition():
    """The schedule bench.py uses for BASELINE configs[3]: reset-aligned shards (no hand-off) with the short tail re-balanced
    over the last two ranks -- exactly one boundary (frame 59) inside a restart unit, served by the hand-off; B1 of all
    local frames, send, then the upsamplers.  Bit-identical to the sequential stream of all 64 frames."""
    from refvsr_amd import shard
    parts = shard.partition_hybrid(64, 8, 9)
    assert parts == [(0, 9), (9, 18), (18, 27), (27, 36), (36, 45), (45, 54), (54, 59), (59, 64)]
    assert [shard.needs_handoff(a, 9) for a, _ in parts] == [False] * 7 + [True]
    got = _run(reset='keep', aligned=False, wavefront='hybrid', world=8, nframes=64, name='config_RefVSR_small_MFID')
    assert sorted(got) == list(range(64))


def test_world8_64_frame_clip_reset9_aligned():
    """The exchange-free alternative on the same clip: shard boundaries snapped to multiples of 9 (units of 9 frames over
    8 ranks, unbalanced tail), no communication at all."""
    got = _run(reset='keep', aligned=True, wavefront=False, world=8, nframes=64, name='config_RefVSR_small_MFID')
    assert sorted(got) == list(range(64))


def test_world8_restart_free_clip_block_cyclic_and_growing_partitions():
    """BASELINE configs[4]'s regime (reset_branch = None: a hand-off at EVERY block boundary) on 8 ranks: the block-cyclic
    partition (blocks of 2 frames dealt round-robin -- ranks own two blocks, the chain visits every rank twice) and the
    growing contiguous shards (shard.partition_chain), through the two-lane hooks of run_wavefront; bit-identical to the
    sequential stream."""
    from refvsr_amd import shard
    blocks = shard.partition_cyclic(26, 8, 2)
    assert len(blocks) == 13 and [r for _, _, r in blocks] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1, 2, 3, 4] and blocks[-1][:2] == (24, 26)
    got = _run(reset=None, aligned=False, wavefront='cyclic', world=8, nframes=26)
    assert sorted(got) == list(range(26))
    assert [b - a for a, b in shard.partition_chain(26, 8)] == sorted(b - a for a, b in shard.partition_chain(26, 8))
    got = _run(reset=None, aligned=False, wavefront='growing', world=8, nframes=26)
    assert sorted(got) == list(range(26))


def _check_exchange(got, infos, nframes, world):
    assert sorted(got) == list(range(nframes))
    prepared = sorted(f for r in infos for f in infos[r]['prepared'])
    assert prepared == list(range(nframes)), 'every context is prepared exactly once across the ranks: %s' % prepared
    assert sum(infos[r]['messages'] for r in infos) == sum(len(infos[r]['imported']) for r in infos) > 0


def test_context_exchange_world2_cyclic_and_reset():
    """run_wavefront(exchange_contexts=True) on two ranks: block-cyclic blocks of 2 (contexts flow in BOTH directions of the one
    rank pair, in frame order) without restarts, and a balanced split with reset_branch = 4 (a restart inside a shard: its
    window needs all five contexts).  Bit-identical to the sequential stream; every context prepared once."""
    got, infos = _run(reset=None, aligned=False, wavefront='exchange_cyclic', world=2, nframes=9)
    _check_exchange(got, infos, 9, 2)
    got, infos = _run(reset=4, aligned=False, wavefront='exchange_balanced', world=2, nframes=9)
    _check_exchange(got, infos, 9, 2)


def test_context_exchange_world8_partitions():
    """The same on 8 ranks: BASELINE configs[3] in miniature (64 frames, reset 9) on the reset-aligned hybrid partition -- the
    restart windows at the block starts import the two contexts before them instead of preparing four extra -- and a
    restart-free 26-frame clip on the block-cyclic and the growing block-cyclic partitions."""
    got, infos = _run(reset='keep', aligned=False, wavefront='exchange_hybrid', world=8, nframes=64, name='config_RefVSR_small_MFID')
    _check_exchange(got, infos, 64, 8)
    for fam in ('exchange_cyclic', 'exchange_growing'):
        got, infos = _run(reset=None, aligned=False, wavefront=fam, world=8, nframes=26)
        _check_exchange(got, infos, 26, 8)


def test_phase_a_groups_world2_and_world8():
    """run_wavefront(group=G) with the context exchange: lane a issues phase A in groups of up to G local windows (Engine.phase_a_group
    on the GPU), the chain's local segments and their upsamplers are issued between the groups (the early chain), blocks behind a
    remote hand-off after lane a.  Two ranks, block-cyclic blocks of 2 with a group size that straddles blocks; eight ranks on
    BASELINE configs[3] in miniature (reset-aligned hybrid: six chains without any hand-off).  Bit-identical to the sequential stream,
    every context prepared once, groups as group_lane_ops forms them."""
    got, infos = _run(reset=None, aligned=False, wavefront='exchange_cyclic@g3', world=2, nframes=9)
    _check_exchange(got, infos, 9, 2)
    assert infos[0]['groups'] == [(0, 1, 4), (5, 8)] and infos[1]['groups'] == [(2, 3, 6), (7,)]
    got, infos = _run(reset='keep', aligned=False, wavefront='exchange_hybrid@g4', world=8, nframes=64, name='config_RefVSR_small_MFID')
    _check_exchange(got, infos, 64, 8)
    assert infos[0]['groups'] == [(0, 1, 2, 3), (4, 5, 6, 7), (8,)] and infos[7]['groups'] == [(59, 60, 61, 62), (63,)]


def test_two_message_handoff():
    """run_wavefront with an executor that offers the two-message hand-off (split_handoff): block-cyclic blocks of 2 on three
    ranks, a hand-off behind every block; the second message is awaited inside the receiver's forward-branch step.  Bit-identical
    to the sequential stream."""
    got = _run(reset=None, aligned=False, wavefront='two_message', world=3, nframes=11)
    assert sorted(got) == list(range(11))
