import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with `-m gpu`)")
    config.addinivalue_line("markers", "slow: a CPU test that takes about a minute (full-size oracle frames); part of the default CPU suite")


@pytest.fixture(scope="session")
def golden64():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "deepsvc_ref_64.npz"))


@pytest.fixture(scope="session")
def golden128():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "deepsvc_ref_128.npz"))


@pytest.fixture(scope="session")
def golden256():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "deepsvc_ref_256.npz"))


@pytest.fixture(scope="session")
def w0_state_dict():
    from deepsvc_amd import synth
    return synth.make_model("W0").state_dict()


@pytest.fixture(scope="session")
def oracle_w0(w0_state_dict):
    from oracle.codec import DeepSVCOracle
    o = DeepSVCOracle(w0_state_dict)
    o.update()
    return o
