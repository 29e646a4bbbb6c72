import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` through gpurun)')
    # the oracle runs on the host: use the cores this process may really use (cgroup quota / affinity), not the
    # whole machine's count -- oversubscribed OpenMP teams make the oracle-based tests many times slower
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    if os.environ.get('PYTEST_XDIST_WORKER_COUNT'):
        n = max(1, n // int(os.environ['PYTEST_XDIST_WORKER_COUNT']))
    torch.set_num_threads(max(1, min(n, 32)))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    return {k: (torch.from_numpy(g[k]) if g[k].ndim > 0 else g[k].item()) for k in g.files}


@pytest.fixture(scope='session')
def small_cfg():
    from refvsr_amd import get_config
    cfg = get_config('p', 'm', 'config_RefVSR_small_L1')
    cfg.frame_num = 5
    return cfg


@pytest.fixture(scope='session')
def small_sd(small_cfg):
    from refvsr_amd import make_state_dict
    return make_state_dict(small_cfg, 1234)


def maxdiff(a, b):
    return float((a.float() - b.float()).abs().max())
