import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import lk_oracle
    lk_oracle.lib()
    return lk_oracle


@pytest.fixture(scope="session")
def engine_lib():
    """liblk_engine.so; built with hipcc if it is not there yet."""
    from correlation_amd import build
    build.build()
    import correlation_amd
    return correlation_amd.load_library()


@pytest.fixture(scope="session")
def speckle512():
    from correlation_amd import speckle
    return speckle.speckle_pair(512, 512, p=(1.3, -0.7, 0.002, 0.0, 0.0, -0.001), seed=7)
