"""torchvision stand-in (oracle import only): vgg19().features layer list."""
from . import models  # noqa: F401
