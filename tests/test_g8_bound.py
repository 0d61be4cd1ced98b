"""The inequality behind the int8 image of the gated columns (oracle/g8_bound_oracle.py), on the CPU: with every rounding going
up, the integer bound of a gated_i8 index never falls below the exact gated score minus the margin of the ungated half -- for
DLR-shaped, outlier, tiny, negative (abs mode) and degenerate inputs, with index-value buckets and with a coarse bucket map."""
import zlib

import numpy as np
import pytest

from dhr_amd import synth
from oracle import g8_bound_oracle as G8
from oracle import gip_oracle as O


def _data(kind, rng, n=600, q=6, d=128, c=64):
    cv, ci = synth.make_dlr(rng, n, d, 10, 30)
    qv, qi = synth.make_dlr(rng, q, d, 3, 8)
    cg, qg = cv.astype(np.float32), qv.astype(np.float32)
    cd = (rng.standard_normal((n, c)) * 0.1).astype(np.float16).astype(np.float32)
    qd = (rng.standard_normal((q, c)) * 0.1).astype(np.float32)
    abs_mode = False
    if kind == "outlier_entry":
        cg[7, 5] = 60000.0
    elif kind == "outlier_columns":
        cg[:, 3] *= 50; cg[:, 77] *= 200
    elif kind == "tiny_gated":
        cg *= 1e-3; qg *= 1e-3
    elif kind == "tiny_ungated":
        cd *= 1e-4; qd *= 1e-4
    elif kind == "huge_ungated":
        cd *= 300; qd *= 300
    elif kind == "negative":
        cg *= np.where(rng.random(cg.shape) < 0.3, -1, 1).astype(np.float32)
        qg *= np.where(rng.random(qg.shape) < 0.3, -1, 1).astype(np.float32)
        abs_mode = True
    elif kind == "negative_queries":
        qg *= np.where(rng.random(qg.shape) < 0.5, -1, 1).astype(np.float32)
    elif kind == "zero_query":
        qg[0] = 0; qd[1] = 0; qg[2] = 0; qd[2] = 0
    elif kind == "no_ungated":
        cd = cd[:, :0]; qd = qd[:, :0]
    elif kind == "fp32_queries":
        qg = (qg * np.float32(0.3)).astype(np.float32); qd = (qd * np.float32(0.3)).astype(np.float32)
    return cg.astype(np.float16).astype(np.float32), ci, cd, qg, qi, qd, abs_mode


KINDS = ["dlr", "outlier_entry", "outlier_columns", "tiny_gated", "tiny_ungated", "huge_ungated", "negative", "negative_queries",
         "zero_query", "no_ungated", "fp32_queries"]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("coarse", [False, True])
def test_g8_bound_is_an_upper_bound(kind, coarse):
    rng = np.random.default_rng(zlib.crc32((kind + str(coarse)).encode()))
    cg, ci, cd, qg, qi, qd, abs_mode = _data(kind, rng)
    bq, bd = (qi % 2, ci % 2) if coarse else (qi, ci)          # two buckets per slice (the device's maps are some such function of the index value)
    U, margin = G8.bound_scores(qg, qi, qd, cg, ci, cd, bq, bd, abs_mode)
    cv = np.concatenate([cg, cd], axis=1)
    for q in range(qg.shape[0]):
        ex = O.gip_scores_f64(np.concatenate([qg[q], qd[q]]), qi[q], cv, ci)
        head = (U[q] - (ex - margin[q])).min()
        assert head >= -1e-9 * max(1.0, np.abs(ex).max()), (kind, q, head, margin[q])


def test_g8_units_meet_in_one_integer_sum():
    """Gated unit = 2^shift ungated units, the shift never lets 127 * 127 * d_dlr * 2^shift leave int32, int8 levels stay <= 127."""
    rng = np.random.default_rng(5)
    for kind in KINDS:
        cg, ci, cd, qg, qi, qd, abs_mode = _data(kind, rng)
        step, w, s_ref = G8.corpus_steps(cg)
        assert G8.corpus_image(cg, step).max() <= 127
        has_u = cd.shape[1] > 0
        if has_u:
            from oracle import i8_bound_oracle as I8
            d8u, cs, sc, ec, nc = I8.corpus_image(cd)
        for q in range(qg.shape[0]):
            r = G8.query_units(qg[q], qd[q] if has_u else None, w, s_ref, cs if has_u else None, sc if has_u else 0.0, abs_mode)
            assert r["q8"].max() <= 255 and r["q8"].min() >= 0
            assert 0 <= r["shift"] <= G8.max_shift(qg.shape[1])
            assert abs(r["u"] - r["u_f"] * 2.0 ** r["shift"]) <= 1e-12 * r["u"]
            assert 255.0 * 127.0 * 32 * ((qg.shape[1] + 31) // 32) * 2.0 ** r["shift"] <= 2.0 ** 30
            if r["q8u"] is not None:
                assert np.abs(r["q8u"]).max() <= 127
