"""Same import path as the reference's net/modules.py (conv_block, up_conv)."""
from rpnet_amd.modules import conv_block, up_conv  # noqa: F401
