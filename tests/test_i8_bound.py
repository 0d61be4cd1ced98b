"""The inequality behind the int8 image of the ungated columns (oracle/i8_bound_oracle.py), on the CPU: for random, outlier-column,
spiky and tiny inputs the error of the int8 inner product never exceeds the margin the filter subtracts from its thresholds."""
import zlib

import numpy as np
import pytest

from oracle import i8_bound_oracle as B


@pytest.mark.parametrize("kind", ["gauss", "outlier_columns", "spiky_rows", "one_hot_query", "tiny", "mixed_scale"])
def test_int8_error_within_margin(kind):
    rng = np.random.default_rng(zlib.crc32(kind.encode()))
    n, c = 4000, 128
    d = (rng.standard_normal((n, c)) * 0.1).astype(np.float16).astype(np.float32)
    qs = (rng.standard_normal((6, c)) * 0.1).astype(np.float32)
    if kind == "outlier_columns":
        d[:, 3] *= 50; d[:, 77] *= 30
    elif kind == "spiky_rows":
        d[:] = 0
        d[np.arange(n), rng.integers(0, c, n)] = 40.0
        d[::7] = 40.0
    elif kind == "one_hot_query":
        qs[:] = 0
        qs[np.arange(6), rng.integers(0, c, 6)] = 2.0
    elif kind == "tiny":
        d *= 1e-3; qs *= 1e-3
    elif kind == "mixed_scale":
        d *= np.exp(rng.standard_normal(c) * 2)[None, :].astype(np.float32)
    ratios = []
    for q in qs:
        err, margin = B.bound_error_and_margin(q, d)
        assert err <= margin * (1 + 1e-9) + 1e-12, (kind, err, margin)
        ratios.append(err / max(margin, 1e-30))
    assert max(ratios) <= 1.0


def test_column_steps_beat_one_global_step_on_outlier_columns():
    """What the per-column steps are for: with two large columns a single corpus-wide step would leave the other columns a handful of
    int8 levels; the margin with column steps is several times smaller."""
    rng = np.random.default_rng(3)
    d = (rng.standard_normal((3000, 128)) * 0.1).astype(np.float32)
    d[:, 5] *= 50; d[:, 70] *= 30
    q = (rng.standard_normal(128) * 0.1).astype(np.float32)
    _, margin_cols = B.bound_error_and_margin(q, d)
    sc = np.abs(d).max() / 127.0
    d8 = B.quant(d, sc)
    am = np.abs(q).max()
    sq = am / 127.0
    q8 = B.quant(q, sq)
    margin_global = np.linalg.norm(q) * np.linalg.norm(d - sc * d8, axis=1).max() + np.linalg.norm(q - sq * q8) * np.linalg.norm(sc * d8, axis=1).max()
    assert margin_cols < 0.5 * margin_global
