"""Per-model config module, importable as `configs.config_RefVSR_IR_L1` is in the reference
(/root/reference/configs/config_RefVSR_IR_L1.py:8)."""
from refvsr_amd.config import get_config as _get


def get_config(project='', mode='', config='config_RefVSR_IR_L1', data='', LRS='', batch_size=8):
    return _get(project, mode, config or 'config_RefVSR_IR_L1', data, LRS, batch_size)
