"""dhr_write_trec (host code of the C ABI) against the Python writer that restates gip_retrieval.py:333-342, byte for byte, and the
score formatting against Python's own float formatting.  No GPU needed: the library loads, the writer touches no device."""
import math
import os

import numpy as np
import pytest

from dhr_amd import _lib
from dhr_amd.retrieval import gip_retrieval as G


def _lib_or_skip():
    try:
        return _lib.load()
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"library not loadable here: {e}")


def test_format_float_equals_python():
    import ctypes as C
    lib = _lib_or_skip()
    rng = np.random.default_rng(1)
    vals = [0.0, -0.0, 1.0, -1.5, 0.1, 1e-4, 9.999e-5, 1e-5, 1.5e-7, 123456789.0, 1e15, 9999999999999998.0, 1e16, 1.2345e22, 3.4028234663852886e38,
            float("inf"), float("-inf"), float("nan"), 5e-324, 2.2250738585072014e-308, 29.929176330566406, 100.0, 1e3, 65504.0]
    vals += [float(np.float32(x)) for x in rng.standard_normal(3000) * 10.0 ** rng.integers(-8, 9, 3000)]
    vals += [float(x) for x in rng.random(2000)]
    buf = C.create_string_buffer(64)
    for v in vals:
        n = lib.dhr_format_float(v, buf, 64)
        assert n > 0
        got = buf.value.decode()
        assert got == "{}".format(v), (v, got, "{}".format(v))
        if not math.isnan(v):
            assert float(got) == v


@pytest.mark.parametrize("seed", [0, 1])
def test_write_trec_equals_python_writer(tmp_path, seed):
    _lib_or_skip()
    rng = np.random.default_rng(seed)
    nd, nq, k = 5000, 37, 50
    docids = ["d%d" % i for i in range(nd)]
    docids[7] = "MARCO_7é"                               # multi-byte id
    qids = ["q%d" % i for i in range(nq)]
    qids[3] = docids[11]                                 # a query whose own document is in the corpus: skipped when retrieved (:340)
    base = 100                                           # global rows of a shard starting at 100
    rows = rng.integers(0, nd, (nq, k)).astype(np.int64) + base
    rows[3, 5] = 11 + base
    rows[5, 40:] = -1                                    # a short list
    rows[6, :] = -1                                      # an empty one
    scores = (rng.standard_normal((nq, k)) * 10.0 ** rng.integers(-6, 7, (nq, k))).astype(np.float32)
    scores[2, 0] = np.float32("inf"); scores[2, 1] = np.float32(0.0); scores[2, 2] = np.float32(1e-5); scores[2, 3] = np.float32(1e16)
    ref = tmp_path / "ref.trec"
    with open(ref, "w") as f:
        G.write_trec(f, *G._to_dicts(qids, rows, scores, base), docids, "h2oloo")
    out = tmp_path / "out.trec"
    n = G.write_trec_native(str(out), qids, rows, scores, base, docids, "h2oloo")
    a, b = open(ref, "rb").read(), open(out, "rb").read()
    assert a == b
    assert n == a.count(b"\n") and n == nq * k - 1 - 10 - k
    # append mode and the fallback signal for ids that are not strings
    assert G.write_trec_native(str(out), qids, rows, scores, base, docids, "h2oloo", append=True) == n
    assert open(out, "rb").read() == a + a
    assert G.write_trec_native(str(out), qids, rows, scores, base, list(range(nd)), "h2oloo") is None
    rows[0, 0] = nd + base + 5
    with pytest.raises(_lib.DhrError):
        G.write_trec_native(str(out), qids, rows, scores, base, docids, "h2oloo")
