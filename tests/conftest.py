import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "drone_golden.npz"))


@pytest.fixture(scope="session")
def att_golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "attention_golden.npz"))


@pytest.fixture(scope="session")
def shapes():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "shapes.json")) as f:
        return json.load(f)
