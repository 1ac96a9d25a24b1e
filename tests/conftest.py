import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["06-leaves-constant-rate", "10-leaves-autocorrelated-rate", "12-leaves-variable-rate", "24-leaves-braces",
            "25-leaves-bastien"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return {name: dict(np.load(os.path.join(GOLDEN, name + ".npz"))) for name in FIXTURES}


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible: the HIP path must run, there is no fallback")
    import mcmc_date_amd

    mcmc_date_amd._capi.lib()
    return torch.device("cuda:0")


class _Knobs:
    """The library's test / tuning knobs (mcd_set_option) with monkeypatch's setenv / delenv spelling: the names are the environment
    variables of rounds 1-3, which the library no longer reads after it has been loaded.  Every knob touched goes back to its default at
    the end of the test."""

    def __init__(self):
        self.touched = set()

    def setenv(self, name, value):
        import mcmc_date_amd as M

        M.set_option(name, value)
        self.touched.add(name)

    def delenv(self, name, raising=True):
        import mcmc_date_amd as M

        M.set_option(name, None)

    def restore(self):
        import mcmc_date_amd as M

        for name in self.touched:
            M.set_option(name, None)


@pytest.fixture
def knobs():
    k = _Knobs()
    yield k
    k.restore()
