"""mmcv.cnn stand-in: ConvModule == nn.Conv2d(bias=True) [+ activation], children named
`conv` / `activate` (public mmcv semantics for norm_cfg=None)."""
import torch.nn as nn


class _Registry:
    def register_module(self, *a, **k):
        def deco(cls):
            return cls
        return deco


CONV_LAYERS = _Registry()


def build_activation_layer(cfg):
    t = cfg['type']
    if t == 'ReLU':
        return nn.ReLU(inplace=False)
    if t == 'LeakyReLU':
        return nn.LeakyReLU(cfg.get('negative_slope', 0.01))
    raise NotImplementedError(t)


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        assert norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            self.activate = build_activation_layer(act_cfg)

    def forward(self, x):
        x = self.conv(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    nn.init.xavier_normal_(module.weight, gain=gain)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)
