"""Container-only (skipped where /root/reference is absent, i.e. on the GPU box): the reference's OWN shell resolves this build through
its arch-plugin hook -- `importlib.import_module('models.archs.' + config.network).Network(config)`, models/SRNet.py:20-21, selected by
`--network` (run.py:364,384) -- with the 3-line `models/archs/RefVSR_MI355X.py` INTEGRATION.md section 1 prints, the reference's own
config object, and a `module.`-prefixed checkpoint loaded through the reference's own CKPT_Manager (ckpt_manager.py:50-56).  No GPU
needed: construction, the state-dict contract and checkpoint loading are host work.  Nothing is written into the reference tree: the
one new file lives in an overlay directory that Python merges into the (namespace) package `models.archs`."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

CHILD = r'''
import importlib, os, re, sys, tempfile
ROOT, REF = sys.argv[1], sys.argv[2]
sys.dont_write_bytecode = True
doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
m = re.search(r"```python\n# models/archs/RefVSR_MI355X\.py[^\n]*\n(.*?)```", doc, re.S)
assert m, 'INTEGRATION.md section 1 no longer prints the plug-in file'
plugin = m.group(1)
assert len([l for l in plugin.strip().splitlines() if l.strip()]) <= 3
plugin = plugin.replace('/path/to/this/repo', ROOT)
name = sys.argv[3]
if '_IR_' in name:             # INTEGRATION.md section 1: the RefVSR_IR family imports the same class from refvsr_amd.archs.RefVSR_IR
    assert 'refvsr_amd.model' in plugin
    plugin = plugin.replace('refvsr_amd.model', 'refvsr_amd.archs.RefVSR_IR')
over = tempfile.mkdtemp()
os.makedirs(os.path.join(over, 'models', 'archs'))
open(os.path.join(over, 'models', 'archs', 'RefVSR_MI355X.py'), 'w').write(plugin)
# the reference tree first (its own modules win), the third-party stand-ins of tools/ref_shims, then the overlay with the ONE new file
sys.path[:0] = [REF, os.path.join(ROOT, 'tools', 'ref_shims'), over]
import torch
cfg = importlib.import_module('configs.' + name).get_config('p', 'm', name)          # the reference's own config object
cfg.cuda, cfg.device, cfg.dist = False, 'cpu', False
cfg.network_ref = cfg.network
from models.SRNet import SRNet                                                      # the reference's own shell
ref_net = SRNet(cfg).eval()
cfg2 = importlib.import_module('configs.' + name).get_config('p', 'm', name)
cfg2.cuda, cfg2.device, cfg2.dist = False, 'cpu', False
cfg2.network = 'RefVSR_MI355X'                                                      # run.py:384: config.network = args.network
net = SRNet(cfg2).eval()
import refvsr_amd.model as mine
assert type(net.Network) is mine.Network or isinstance(net.Network, mine.Network), type(net.Network)
assert hasattr(net.Network, 'FlowNet')                                              # models/SRNet.py:44
net.init()                                                                          # wi / win are None in every config: a no-op
ks_ref, ks = ref_net.state_dict(), net.state_dict()
assert list(ks_ref.keys()) == list(ks.keys()) or set(ks_ref.keys()) == set(ks.keys()), set(ks_ref.keys()) ^ set(ks.keys())
for k, v in ks_ref.items():
    assert tuple(v.shape) == tuple(ks[k].shape) and v.dtype == ks[k].dtype, (k, v.shape, ks[k].shape)
# a released checkpoint = the DataParallel-wrapped SRNet's state dict: every key carries a leading `module.`
g = torch.Generator().manual_seed(5)
ck = {'module.' + k: torch.randn(v.shape, generator=g).to(v.dtype) if v.dtype.is_floating_point else v.clone() for k, v in ks_ref.items()}
path = os.path.join(over, 'RefVSR_test.pytorch')
torch.save(ck, path)
from ckpt_manager import CKPT_Manager                                               # the reference's own loader
mgr = CKPT_Manager(over, 'm', False, False)
res, fname = mgr.load_ckpt(net, abs_name=path)
assert fname == 'RefVSR_test.pytorch'
assert not res.missing_keys and not res.unexpected_keys, (res.missing_keys[:3], res.unexpected_keys[:3])
res_ref, _ = mgr.load_ckpt(ref_net, abs_name=path)
assert not res_ref.missing_keys and not res_ref.unexpected_keys
for k, v in net.state_dict().items():
    assert torch.equal(v, ck['module.' + k]) and torch.equal(v, ref_net.state_dict()[k]), k
# the call contract without a GPU: the build refuses CPU tensors loudly (no silent fallback), like any missing-extension case
x = torch.rand(1, 3, 3, 16, 16)
try:
    net(x, x, True)
    raise SystemExit('a CPU call must raise')
except RuntimeError as e:
    assert 'GPU' in str(e) or 'librefvsr_hip' in str(e), e
print('PLUGIN_OK', name, len(ks))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree only exists in the build container')
@pytest.mark.parametrize('name', ['config_RefVSR_small_L1', 'config_RefVSR_MFID_8K', 'config_RefVSR_IR_L1'])
def test_reference_shell_resolves_the_plugin_and_loads_a_module_prefixed_checkpoint(name):
    r = subprocess.run([sys.executable, '-c', CHILD, ROOT, REF, name], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'PLUGIN_OK ' + name in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
