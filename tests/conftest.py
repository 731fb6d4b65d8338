import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
EMU_BUILD = os.path.join(ROOT, "tests", "emu", "_build")
PRODUCT_TESTS = os.path.join(ROOT, "build", "tests")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is dominated by independent emulator runs, each a process of its own on one core: spread them over
    a few pytest-xdist workers when the plugin is installed and the caller did not choose a worker count (every multi-process test
    uses a rendezvous port of its own).  GPU runs stay in one process: one GPU, one user at a time.
    xdist WORKERS run this hook as well (xdist/remote.py calls pytest_cmdline_main on the worker's config): a worker that set
    numprocesses would become a controller of its own and spawn workers that do the same.  Three independent guards: the worker's
    config carries `workerinput`, its environment carries PYTEST_XDIST_WORKER, and every descendant of the process that switched the
    workers on inherits OMMHIP_XDIST_PARENT."""
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER") or os.environ.get("OMMHIP_XDIST_PARENT"):
        return None
    markexpr = getattr(config.option, "markexpr", "") or ""
    if "not gpu" not in markexpr or not config.pluginmanager.hasplugin("xdist") or (os.cpu_count() or 1) < 8:
        return None
    if getattr(config.option, "numprocesses", None) is None and not getattr(config.option, "collectonly", False) and getattr(config.option, "dist", "no") == "no":
        os.environ["OMMHIP_XDIST_PARENT"] = str(os.getpid())
        config.option.numprocesses = 3
        config.option.dist = "load"
        config.option.tx = ["popen"] * 3
    return None


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
