import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import orclib
    orclib.lib()
    return orclib


@pytest.fixture(scope="session")
def samples_seeded(orc):
    return orc.gen_samples()


@pytest.fixture(scope="session")
def samples_half(orc):
    return orc.const_samples(0.5)
