import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real AMD GPU (run with -m gpu on the MI355X box)")


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


_SWITCHES = ("ISOCHRONES_AMD_PATH", "ISOCHRONES_AMD_SAMPLER", "ISOCHRONES_AMD_QUANTILES", "ISOCHRONES_AMD_HOST_SYNC",
             "ISOCHRONES_AMD_TREE_RUNTIME_LEAVES", "ISOCHRONES_AMD_DENSE_THREADS", "ISOCHRONES_AMD_DENSE_STDP", "ISOCHRONES_AMD_STD_PRIORS", "ISOCHRONES_AMD_STAR_LANES")


@pytest.fixture(autouse=True)
def _kernel_switches_do_not_leak():
    """The library reads its kernel-selection switches from the environment at call time: a test that leaves one
    set changes what every later test measures.  Restore them, and say which test leaked."""
    before = {k: os.environ.get(k) for k in _SWITCHES}
    yield
    leaked = {k: os.environ.get(k) for k in _SWITCHES if os.environ.get(k) != before[k]}
    for k, v in before.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    assert not leaked, "test left kernel switches set: %r" % leaked
