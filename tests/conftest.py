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
