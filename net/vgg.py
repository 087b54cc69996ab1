"""Same import path as the reference's net/vgg.py (Encoder)."""
from rpnet_amd.modules import Encoder  # noqa: F401
