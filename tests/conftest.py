import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "vcs-h264_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG, os.path.join(PKG, "InterframeCompression")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("TQDM_DISABLE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/liboracle.so), built on demand."""
    from oracle import oracle as o
    o.build()
    o.lib()
    return o
