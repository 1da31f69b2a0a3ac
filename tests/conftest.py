import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def suzanne():
    import numpy as np

    d = np.load(os.path.join(ROOT, "tests", "golden", "suzanne.npz"))
    return d["vertices"].astype(np.float32), d["indices"].astype(np.uint32)
