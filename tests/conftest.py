import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The int8 image of the gated columns is the library's default for shards of at least 1 M rows only; the parity tests are small, and they
# must exercise it (it is what the headline configuration runs): force it on unless a test sets the variable itself (monkeypatch).
os.environ.setdefault("DHR_GATED_I8", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from tests.util import Golden
    return Golden()
