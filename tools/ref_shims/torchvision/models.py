import torch.nn as nn

_CFG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class _VGG:
    def __init__(self):
        layers, c = [], 3
        for v in _CFG_E:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(c, v, kernel_size=3, padding=1), nn.ReLU(inplace=False)]
                c = v
        self.features = nn.Sequential(*layers)


def vgg19(pretrained=False, **kw):
    return _VGG()
