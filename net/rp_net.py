"""Same import path as the reference's net/rp_net.py; implementation in rpnet_amd (HIP kernels)."""
from rpnet_amd.modules import RP_Net, ContextCorrelationEncoder  # noqa: F401
from rpnet_amd.functional import dice_ce  # noqa: F401
