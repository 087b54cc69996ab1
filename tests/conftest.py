import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
    return load


@pytest.fixture(autouse=True)
def _library_defaults():
    """Every test starts from, and leaves behind, the library's default switches (convolution arithmetic f16x2, fp16
    threshold, tile choice, weight-gradient stream): a test that changes one cannot leak it into the next."""
    import rpnet_amd.functional as RF
    import rpnet_amd.modules as RM
    saved = (RF.conv_math(), RM._F16_MIN_PIXELS, RF._ASYNC["on"], dict(RF.TUNE))
    yield
    RF.set_conv_math(saved[0])
    RM._F16_MIN_PIXELS = saved[1]
    RF.set_async_wgrad(saved[2])
    RF.TUNE.clear()
    RF.TUNE.update(saved[3])
