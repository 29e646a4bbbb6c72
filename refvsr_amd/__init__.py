"""refvsr_amd -- MI355X (gfx950) native RefVSR inference hot path.

    from refvsr_amd import SRNet, get_config
    cfg = get_config('proj', 'mode', 'config_RefVSR_small_L1'); cfg.frame_num = 5
    net = SRNet(cfg).cuda().eval(); net.load_state_dict(ckpt)
    out = net(lr_window, ref_window, is_first_frame)['result']
"""
from .config import CONFIG_NAMES, get_config, set_data_path, set_scale  # noqa: F401
from .weights import make_state_dict, state_spec  # noqa: F401


def __getattr__(name):
    if name in ('SRNet', 'Network'):
        from . import model
        return getattr(model, name)
    raise AttributeError(name)
