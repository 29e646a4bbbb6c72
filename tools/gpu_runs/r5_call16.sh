#!/bin/bash
# round 5, call 16: n > 1 samples of a call as multi-map launches (Engine.forward_multi): bit-identity + the API contract test + rate
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r5_call16.log
: > $L
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q --no-header -p no:cacheprovider --timeout 400 --timeout-method=thread -x -k "batch_samples or batch_and_api or frame_groups_are or pipelined_mode" 2>&1 | tail -8 | tee -a $L
echo "== n = 4 clips per call vs 4 x n = 1 ==" | tee -a $L
timeout 300 python - <<'PY' 2>&1 | grep "^multi" | tee -a $L
import sys, time, torch
sys.path.insert(0, '.')
from refvsr_amd import SRNet, get_config, make_state_dict
from refvsr_amd.synth import make_clip, window_indices
dev = torch.device('cuda:0')
cfg = get_config('p', 'm', 'config_RefVSR_small_L1'); cfg.frame_num = 5
sd = make_state_dict(cfg, 1234)
nfr, t, n = 25, 5, 4
clips = [make_clip(nfr, 270, 480, seed=b, want_gt=False) for b in range(n)]
lr = torch.stack([c[0] for c in clips], 0).to(dev); rf = torch.stack([c[1] for c in clips], 0).to(dev)
wins = [window_indices(f, nfr, t) for f in range(nfr)]
xs = [lr[:, w].contiguous() for w in wins]; rs = [rf[:, w].contiguous() for w in wins]
torch.cuda.synchronize()
for label, nn in (('n=4 per call (multi-map)', 4), ('n=1 per call (one clip)', 1)):
    net = SRNet(cfg).to(dev).eval(); net.load_state_dict(sd); net.Network.set_pipelined(True)
    best = 0.0
    for rep in range(4):
        net.Network.reset()
        for f in range(5):
            net(xs[f][:nn], rs[f][:nn], f == 0, frame_ids=wins[f], input_ready='materialised')
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f in range(5, nfr):
            o = net(xs[f][:nn], rs[f][:nn], f == 0, frame_ids=wins[f], input_ready='materialised')['result']
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        best = max(best, nn * (nfr - 5) / el)
    print('multi %-28s %.1f output frames/s' % (label, best))
PY
