import os
import sys

import pytest

# torch bundles its own HIP runtime (same soname as /opt/rocm's).  Whichever copy is loaded first
# serves the whole process, so load torch's before librtx.so pulls in the system one: the
# device-resident entry point is then exercised exactly as bench.py uses it.
try:  # noqa: SIM105
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is plumbing, tests that need it importorskip
    torch = None

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
