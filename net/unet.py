"""Same import path as the reference's net/unet.py (U_Net, Unet_2D)."""
from rpnet_amd.modules import U_Net, Unet_2D  # noqa: F401
