"""TEST INFRASTRUCTURE ONLY (see oracle/gip_oracle.py's header): numpy restatement of the int8 image of the ungated columns that the
bound GEMM of dhr_amd uses (dhr_amd/csrc/kernels.hip: tile_rows_sparse_kernel, i8_row_err_kernel, query_prep_kernel) and of the
margin that pays for it.  It has no counterpart in the reference (retrieval/gip_retrieval.py computes the inner product in fp32,
:115-126): the search result never depends on it, only the number of rows that reach the exact rescoring does.  What the tests pin
is the inequality the filter relies on,

    | <q, d> - mul * <q8, d8> |  <=  ||q'|| * ec + ||q' - sq q8|| * nc            for every query q and every corpus row d,

with  sc = max |d| / 127,  cs_j = sc * (max_r |d_rj| / max |d|)^(3/4) (column steps),  d8_rj = clamp(rint(d_rj / cs_j)),  w_j = cs_j / sc,
q'_j = q_j w_j,  sq = max_j |q'_j| / 127,  q8 = clamp(rint(q' / sq)),  mul = sc * sq,
ec = max_r || (d_r - cs * d8_r) / w ||,  nc = max_r || sc * d8_r ||.
"""
import numpy as np


def quant(x, step):
    return np.clip(np.rint(x / step), -127, 127)


def corpus_image(d: np.ndarray):
    """d: float [N, C] ungated columns -> (d8 int [N, C], cs [C], sc, ec, nc)."""
    d = d.astype(np.float64)
    colmax = np.abs(d).max(axis=0)
    sc = max(colmax.max() / 127.0, 1e-30)
    cs = sc * np.maximum(np.where(colmax > 0, np.minimum(colmax / (127.0 * sc), 1.0), 1.0) ** 0.75, 1.0 / 1024.0)
    d8 = quant(d, cs[None, :])
    w = cs / sc
    ec = np.linalg.norm((d - cs[None, :] * d8) / w[None, :], axis=1).max()
    nc = np.linalg.norm(sc * d8, axis=1).max()
    return d8, cs, sc, ec, nc


def query_image(q: np.ndarray, cs: np.ndarray, sc: float):
    """q: float [C] -> (q8, sq, ||q'||, ||q' - sq q8||)."""
    qp = q.astype(np.float64) * (cs / sc)
    am = np.abs(qp).max()
    sq = am / 127.0 if am > 0 else 1.0
    q8 = quant(qp, sq)
    return q8, sq, np.linalg.norm(qp), np.linalg.norm(qp - sq * q8)


def bound_error_and_margin(q: np.ndarray, d: np.ndarray):
    """-> (max_r |<q,d_r> - mul <q8,d8_r>|, margin) for one query against a corpus block."""
    d8, cs, sc, ec, nc = corpus_image(d)
    q8, sq, qn, qe = query_image(q, cs, sc)
    approx = (sc * sq) * (d8 @ q8)
    exact = d.astype(np.float64) @ q.astype(np.float64)
    return float(np.abs(exact - approx).max()), float(qn * ec + qe * nc)
