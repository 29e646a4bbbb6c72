"""Drop-in model shell: `SRNet(config)` with the reference's constructor / forward / state_dict
surface (/root/reference/models/SRNet.py:11-61, models/archs/RefVSR.py:14-101,151-325), executing on
the HIP engine (refvsr_amd/engine.py).

  * `SRNet(config).Network` is resolved through `config.network` exactly like the reference's
    importlib plugin hook (SRNet.py:20-21): `refvsr_amd.archs.<config.network>.Network(config)`.
  * Parameters are registered under the reference's state-dict key names (refvsr_amd/weights.py), so
    `load_state_dict` accepts released checkpoints (with or without the DataParallel `module.`
    prefix, ckpt_manager.py:50-56) and `state_dict()` round-trips.
  * The recurrent forward state lives on the module between calls, like the reference
    (RefVSR.py:96-101,279-283): one module instance == one stream of consecutive frames, not
    thread-safe, not re-entrant.
  * There is NO CPU fallback: inputs must be CUDA (HIP) tensors and the HIP library must be built.
"""
import collections
import importlib

import torch
import torch.nn as nn

from . import hip
from .engine import Engine, Weights
from .weights import state_spec, strip_module_prefix


def _register(root, dotted, param):
    mod = root
    parts = dotted.split('.')
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, nn.Module())
        mod = getattr(mod, p)
    mod.register_parameter(parts[-1], param)


class _FlowNetHandle(nn.Module):
    """Placeholder carrying `load_ckpt` so that `SRNet.init()` keeps working (SRNet.py:44-45)."""

    def load_ckpt(self, pretrained):
        sd = torch.load(pretrained, map_location='cpu')
        sd = sd.get('state_dict', sd)
        own = dict(self.named_parameters())
        with torch.no_grad():
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v)


class Network(nn.Module):
    """HIP implementation of models/archs/RefVSR.py:Network (inference path)."""
    _engine_cls = Engine
    _weights_cls = Weights
    _family = 'RefVSR'                  # which state-dict contract this class carries, whatever name config.network has

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.rank = torch.distributed.get_rank() if config.dist else -1
        self.scale = config.scale
        self.flag_HD_in = config.flag_HD_in
        self.mid_channels = config.mid_channels
        self.add_module('FlowNet', _FlowNetHandle())
        for name, shape in state_spec(config, self._family).items():
            assert name.startswith('Network.')
            p = nn.Parameter(torch.zeros(shape), requires_grad=False)
            _register(self, name[len('Network.'):], p)
        self._plist = None
        self._packed = None
        self._packed_key = None
        self._engines = []

    # -- weights ------------------------------------------------------------------------------
    def _weights_key(self):
        # cheap change detector (a full walk over named_parameters() costs > 1 ms per call): cached parameter list,
        # summed in-place version counters, storage address and device of the first / last tensor
        if self._plist is None:
            self._plist = list(self.parameters())
        pl = self._plist
        return (len(pl), sum(p._version for p in pl), pl[0].data_ptr(), pl[-1].data_ptr(), pl[0].device, pl[0].dtype)

    def _apply(self, fn, *a, **k):               # .to() / .cuda() / .half() replace parameter storage
        self._plist = None
        return super()._apply(fn, *a, **k)

    def _weights(self, device):
        key = (self._weights_key(), str(device))
        if self._packed is None or self._packed_key != key:
            sd = collections.OrderedDict(('Network.' + k, v.detach()) for k, v in self.named_parameters())
            self._packed = self._weights_cls(self.config, sd, device)
            self._packed_key = key
            for e in self._engines:
                # everything an engine caches (per-frame matching / encodings / flows, the forward-branch state) was
                # computed with the old weights
                e.W = self._packed
                e.reset_state()
        return self._packed

    def reset(self):
        """Forget the recurrent state (start a new clip)."""
        for e in self._engines:
            e.reset_state()

    # mirrors the attributes of the reference module that callers / tests look at
    @property
    def frame_itr_num(self):
        return self._engines[0].frame_itr_num if self._engines else 0

    def engine(self, n=0):
        return self._engines[n]

    def set_pipelined(self, on=True):
        """Cross-call stream pipelining (see Engine.set_pipelined for the contract)."""
        self.config.pipelined = bool(on)
        for e in self._engines:
            e.set_pipelined(on)

    def ensure_engines(self, n, device):
        W = self._weights(device)
        while len(self._engines) < n:
            self._engines.append(self._engine_cls(self.config, W))
        return self._engines

    def forward(self, lrs, refs, is_first_frame, is_log=False, is_train=False, frame_ids=None, input_ready=None):
        """Same contract as RefVSR.py:151: lrs, refs [n,t,3,h,w] in [0,1]; returns OrderedDict with
        'result' [n,3,4h,4w] (+ 'eval_vis' when is_log and config.save_sample).
        frame_ids / input_ready: extensions (window cache keyed by ids; when the inputs are final -- Engine.set_pipelined)."""
        if is_train:
            raise NotImplementedError('refvsr_amd implements the inference path only (is_train=False)')
        hip.lib()                                              # fail loudly if the extension is missing
        if not lrs.is_cuda:
            raise RuntimeError('refvsr_amd.Network runs on the GPU only (got a %s tensor); there is no CPU path'
                               % lrs.device)
        n = lrs.shape[0]
        self.ensure_engines(n, lrs.device)
        if input_ready is not None and any(e.takes_pipelined_path(frame_ids, bool(is_log)) for e in self._engines[:n]):
            # the engine's internal streams will read the inputs when `input_ready` says so, not in the caller's stream order: a
            # dtype / layout conversion here would be pending work on the caller's stream that nothing waits for -- refused
            # instead of raced against.  (input_ready=None waits for the caller's stream, conversions included; calls that take
            # the sequential path -- is_log, cache or overlap off -- run in the caller's stream order anyway.)
            for t_ in (lrs, refs):
                if t_.dtype != torch.float32 or not t_.is_contiguous():
                    raise RuntimeError('pipelined mode with input_ready= needs contiguous float32 inputs (got %s, contiguous=%s): '
                                       'convert before the producer signals, or pass input_ready=None' % (t_.dtype, t_.is_contiguous()))
        lrs = lrs.float().contiguous()
        refs = refs.float().contiguous()
        want_vis = bool(is_log and self.config.save_sample)
        results, vis_all = [], []
        dbg_all = []
        if (n > 1 and frame_ids is not None and not is_log and
                all(e.takes_pipelined_path(frame_ids, False) and e.group_ok() for e in self._engines[:n])):
            # round 5: the n samples are n independent streams over identical weights -- in steady state their forward-branch steps and
            # backward branches run as multi-map launches (Engine.forward_multi); same results as the loop below, bit for bit
            res = self._engine_cls.forward_multi(self._engines[:n], lrs, refs, bool(is_first_frame),
                                                 [[(b, f) for f in frame_ids] for b in range(n)], input_ready)
            outs = collections.OrderedDict()
            outs['result'] = torch.stack(res, 0)
            return outs
        for b in range(n):
            out, vis = self._engines[b].forward(lrs[b], refs[b], bool(is_first_frame), want_vis,
                                                None if frame_ids is None else [(b, f) for f in frame_ids], want_log=bool(is_log),
                                                input_ready=input_ready)
            if is_log:
                vis, dbg = vis
                dbg_all.append(dbg)
            results.append(out)
            vis_all.append(vis)
        outs = collections.OrderedDict()
        if is_log:                                             # RefVSR.py:162-164,219-221,262-263,301-316
            outs['vis'] = collections.OrderedDict((k, torch.stack([d[k] for d in dbg_all], 0)) for k in dbg_all[0])
        outs['result'] = results[0].unsqueeze(0) if n == 1 else torch.stack(results, 0)     # (n == 1: a view, no 25 MB copy)
        if want_vis and vis_all[0] is not None:                # (RefVSR_IR has no 'eval_vis')
            ev = collections.OrderedDict()
            for k in vis_all[0]:
                ev[k] = torch.stack([v[k] for v in vis_all], 0)
            outs['eval_vis'] = ev
        return outs

    def forward_group(self, lrs, refs, frame_ids, is_first_frame=False, input_ready=None):
        """B CONSECUTIVE windows of one stream in one call (extension; see Engine.forward_group): lrs, refs [B,t,3,h,w] -- window b
        is what the b-th of B consecutive forward() calls would get as its n = 1 input -- frame_ids: B lists of t ids.  Returns
        OrderedDict{'result': tuple of B tensors [1,3,sh,sw]}, bit-identical to the B calls; the backward branches and the
        upsamplers' inputs of the B frames run as multi-map launches, the forward-branch steps frame by frame."""
        hip.lib()
        if not lrs.is_cuda:
            raise RuntimeError('refvsr_amd.Network runs on the GPU only (got a %s tensor); there is no CPU path' % lrs.device)
        for t_ in (lrs, refs):
            if t_.dtype != torch.float32 or not t_.is_contiguous() or t_.dim() != 5:
                raise RuntimeError('forward_group needs contiguous float32 [B,t,3,h,w] inputs (got %s, contiguous=%s)' % (t_.dtype, t_.is_contiguous()))
        assert len(frame_ids) == lrs.shape[0] and lrs.shape == refs.shape
        eng = self.ensure_engines(1, lrs.device)[0]
        wins = [(lrs[b], refs[b], [(0, f) for f in frame_ids[b]]) for b in range(lrs.shape[0])]
        res = eng.forward_group(wins, bool(is_first_frame), input_ready)
        outs = collections.OrderedDict()
        outs['result'] = tuple(r.unsqueeze(0) for r in res)
        return outs

    # ---- two-phase forward for the multi-GPU wavefront (refvsr_amd/shard.py:run_wavefront; not in the reference) ----
    def phase_a(self, lrs, refs, frame_ids=None, first_hint=False):
        """State-independent part of forward() (preparation + backward branch) for lrs, refs [n,t,3,h,w]."""
        hip.lib()
        if not lrs.is_cuda:
            raise RuntimeError('refvsr_amd.Network runs on the GPU only (got a %s tensor); there is no CPU path' % lrs.device)
        lrs, refs = lrs.float().contiguous(), refs.float().contiguous()
        self.ensure_engines(lrs.shape[0], lrs.device)
        return [self._engines[b].phase_a(lrs[b], refs[b], None if frame_ids is None else [(b, f) for f in frame_ids],
                                         first_hint) for b in range(lrs.shape[0])]

    def phase_a_group(self, lrs, refs, frame_ids, first_hints=None, streams=None):
        """phase_a of B windows of ONE clip in one pass (round 6; see Engine.phase_a_group): lrs, refs [B,t,3,h,w], frame_ids B lists of t
        ids (or sequences of B window tensors [t,3,h,w]).  Returns B handle lists (one per window, each what phase_a returns for an
        n = 1 call)."""
        hip.lib()
        if not lrs[0].is_cuda:
            raise RuntimeError('refvsr_amd.Network runs on the GPU only (got a %s tensor); there is no CPU path' % lrs[0].device)
        assert len(lrs) == len(refs) == len(frame_ids)
        eng = self.ensure_engines(1, lrs[0].device)[0]
        wins = [(lrs[b].float().contiguous(), refs[b].float().contiguous(), [(0, f) for f in frame_ids[b]]) for b in range(len(lrs))]
        return [[h] for h in eng.phase_a_group(wins, first_hints, streams)]

    def phase_b1(self, handles, is_first_frame):
        """Serial part of phase_b (forward-branch step, carried state); see Engine.phase_b1."""
        for b, h in enumerate(handles):
            self._engines[b].phase_b1(h, bool(is_first_frame))
        return handles

    def phase_b2(self, handles, is_log=False):
        """State-free rest of phase_b (BW/FW fusion + upsampler); same return value as forward()."""
        want_vis = bool(is_log and self.config.save_sample)
        res = [self._engines[b].phase_b2(h, want_vis) for b, h in enumerate(handles)]
        outs = collections.OrderedDict()
        if is_log:
            outs['vis'] = collections.OrderedDict()
        outs['result'] = res[0][0].unsqueeze(0) if len(res) == 1 else torch.stack([r[0] for r in res], 0)
        if want_vis:
            ev = collections.OrderedDict()
            for k in res[0][1]:
                ev[k] = torch.stack([r[1][k] for r in res], 0)
            outs['eval_vis'] = ev
        return outs

    def phase_b(self, handles, is_first_frame, is_log=False, after_state=None):
        """State-dependent rest (forward-branch step + upsampler); same return value as forward().
        after_state: see Engine.phase_b (called for sample 0)."""
        want_vis = bool(is_log and self.config.save_sample)
        res = [self._engines[b].phase_b(h, bool(is_first_frame), want_vis, after_state if b == 0 else None)
               for b, h in enumerate(handles)]
        outs = collections.OrderedDict()
        if is_log:
            outs['vis'] = collections.OrderedDict()
        outs['result'] = res[0][0].unsqueeze(0) if len(res) == 1 else torch.stack([r[0] for r in res], 0)
        if want_vis:
            ev = collections.OrderedDict()
            for k in res[0][1]:
                ev[k] = torch.stack([r[1][k] for r in res], 0)
            outs['eval_vis'] = ev
        return outs


class SRNet(nn.Module):
    """models/SRNet.py:SRNet surface."""

    def __init__(self, config):
        super().__init__()
        self.rank = torch.distributed.get_rank() if config.dist else -1
        self.config = config
        self.device = config.device
        lib = importlib.import_module('refvsr_amd.archs.{}'.format(config.network))
        self.Network = lib.Network(config)
        self.data = collections.OrderedDict()

    def init(self):
        # dead in the reference too (wi / win are None in every config, SRNet.py:40-45)
        if self.config.wi is not None and self.config.win is not None:
            raise NotImplementedError('weight initialisation is a training feature')

    def input_constructor(self, res):
        b, f, c, h, w = res[:]
        imgs = torch.rand(b, f, c, h, w, device=self.device)
        return {'x': imgs, 'ref': imgs}

    def load_state_dict(self, state_dict, strict=True):
        return super().load_state_dict(strip_module_prefix(state_dict), strict=strict)

    def forward(self, x, ref, is_first_frame=True, is_log=False, is_train=False, frame_ids=None, input_ready=None):
        """frame_ids, input_ready (optional extensions, not in the reference): one id per window frame / when the inputs are
        final -- see Engine.forward and Engine.set_pipelined."""
        return self.Network.forward(x, ref, is_first_frame, is_log=is_log, is_train=is_train, frame_ids=frame_ids, input_ready=input_ready)

    def forward_group(self, x, ref, frame_ids, is_first_frame=False, input_ready=None):
        """B consecutive windows of one stream in one call (extension, not in the reference): see Network.forward_group."""
        return self.Network.forward_group(x, ref, frame_ids, is_first_frame=is_first_frame, input_ready=input_ready)
