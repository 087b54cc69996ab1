"""Same import path as the reference's net/model.py:4-7 (`from net.model import model_factory`,
test_rpnet.py:11).  LGCANet_V3 is a different model and out of scope."""
from rpnet_amd.modules import model_factory, RP_Net  # noqa: F401
