from torch.nn.modules.batchnorm import _BatchNorm  # noqa: F401
