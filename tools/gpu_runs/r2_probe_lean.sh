cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
python tools/probe_resblock.py 2>&1 | grep -A10 "== LR (270" | head -11
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q --no-header -p no:cacheprovider -x 2>&1 | tail -2
CONV48_ONLY="S " CONV48_ITERS=100 python tools/bench_conv48.py 2>&1 | grep conv | cut -c1-70
python - <<'P'
import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import torch
from refvsr_amd import ops
from refvsr_amd.packing import pack_conv
import bench_conv48 as b
dev=b.dev; g=torch.Generator().manual_seed(0); C=24
w1=torch.randn(C,C,3,3,generator=g)/(C*9)**0.5
c1=ops.ConvWeights(pack_conv(w1,torch.zeros(C),[C]),dev); c2=ops.ConvWeights(pack_conv(w1.flip(0),torch.zeros(C),[C]),dev)
for name,h,w in (('LR',270,480),('2x',540,960)):
    x=ops.pack_nhwc16(torch.randn(C,h,w,generator=g).to(dev))
    print('resblock %s %.2f us' % (name, b.timeit(lambda: ops.resblock(c1,c2,x,act=0.0), iters=200)))
P
