import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def orc():
    from _libs import OrcLib
    return OrcLib()


@pytest.fixture(scope="session")
def ref():
    from _libs import RefLib
    if not RefLib.available():
        pytest.skip("oracle/_ref/libmiso_ref.so not built (needs /root/reference: make -C oracle ref)")
    return RefLib()
