import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_BUILD = os.path.join(ROOT, "tests", "emu", "_build")
PRODUCT_TESTS = os.path.join(ROOT, "build", "tests")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def max_rel_force_error(f, f_ref):
    """max_i |F_i - Fref_i| / RMS(|Fref|)  -- the force metric of SURVEY.md §8(d)."""
    import numpy as np
    rms = np.sqrt((f_ref ** 2).sum(1).mean())
    return float(np.sqrt(((f - f_ref) ** 2).sum(1)).max() / rms)
