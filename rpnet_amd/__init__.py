"""rpnet_amd — MI355X-native implementation of the RP-Net data-parallel hot path.

HIP kernels + C ABI: rpnet_amd/csrc, include/rpnet_abi.h  (librpnet_hip.so)
host mirror of the reference's nn.Module surface: rpnet_amd.modules
"""
from .modules import RP_Net, U_Net, ContextCorrelationEncoder, conv_block, up_conv, model_factory  # noqa: F401
from .functional import dice_ce  # noqa: F401
