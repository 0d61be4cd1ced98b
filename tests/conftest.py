import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Two images of the gated half exist (DHR_OPT_GATED_I8): fp16 2:4 -- the library's default for every shard below 1 M rows, i.e. for every
# test-sized corpus, config 1 and the mid-size BEIR corpora -- and int8 2:4, the default of the headline configuration.  Nothing is forced
# globally (round 3 forced the int8 image for the whole suite, which left the fp16 production path to three tests): the end-to-end parity,
# sharded, CLI and stress tests take the `gated_image` fixture below and run ONCE PER IMAGE; tests of a single image set the variable
# themselves; everything else runs what the library would choose for its size.


@pytest.fixture(params=["gated_fp16", "gated_i8"])
def gated_image(request, monkeypatch):
    """Runs the test once per image of the gated half (dhr_index_create reads DHR_GATED_I8 on every call)."""
    monkeypatch.setenv("DHR_GATED_I8", "1" if request.param == "gated_i8" else "0")
    return request.param


@pytest.fixture
def force_gated_i8(monkeypatch):
    monkeypatch.setenv("DHR_GATED_I8", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from tests.util import Golden
    return Golden()
