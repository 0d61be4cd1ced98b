"""TEST INFRASTRUCTURE ONLY (see oracle/gip_oracle.py's header): numpy restatement of the int8 image of the GATED columns that
the bound GEMM of a gated_i8 index uses (dhr_amd/csrc/gemm_g8.hip; images built by tile_rows_sparse_kernel and
query_prep_kernel in kernels.hip, column steps by dhr_index_create in api.hip).  It has no counterpart in the reference
(retrieval/gip_retrieval.py:115-126 computes the gated inner product in fp32): the search result never depends on it, only the
number of rows that reach the exact rescoring does.  What the tests pin:

    U(q, d) = u_q * sum_j q8_j d8_j [bucket(q.idx_j) == bucket(d.idx_j)]  +  (int8 image of the ungated columns, i8_bound_oracle)
           >= GIP(q, d) - margin_q                                           for every query and every corpus row,

with  s_ref = max |gated value| / 127,  step_j = s_ref * max((colmax_j / max)^(3/4), 1/1024),  d8 = ceil(|d| / step_j) <= 127,
w_j = step_j / s_ref,  q8_j = ceil(max(q_j, 0) w_j / sqg) <= 255 (8 bits: stored as level - 128, the accumulators start at
128 x the row's sum of d8),  u_q = sqg * s_ref  (every rounding goes UP, so the gated
half needs no margin of its own), and the two halves meet in one integer sum: u_q = 2^shift * (sc * sq).

The device computes these in fp32 with ~1e-6 of deliberate head room; this restatement is the nominal arithmetic in float64, so
a value that sits within 1e-6 of an int8 level boundary may come out one level apart (the tests allow for that).
"""
import numpy as np

from . import i8_bound_oracle as I8


def corpus_steps(cg: np.ndarray):
    """cg: float [N, D] gated values (non-negative, or |.| in abs mode) -> (step [D], w [D], s_ref)."""
    cg = np.abs(cg.astype(np.float64))
    colmax = cg.max(axis=0)
    gmax = max(float(colmax.max()), 0.0) or 1.0
    s_ref = gmax * (1.00001 / 127.0)
    ratio = np.where(colmax > 0, np.minimum(colmax / gmax, 1.0), 1.0)
    f = np.maximum(ratio ** 0.75, 1.0 / 1024.0)
    return s_ref * f, f, s_ref


def corpus_image(cg: np.ndarray, step: np.ndarray):
    return np.minimum(np.ceil(np.abs(cg.astype(np.float64)) / step[None, :] * (1.0 + 1e-6)), 127.0)


def max_shift(d_dlr: int) -> int:
    s = 0
    ts32 = ((d_dlr + 31) // 32) * 32
    while s < 7 and 255.0 * 127.0 * ts32 * (2 << s) <= 2.0 ** 30:
        s += 1
    return s


def query_units(qg: np.ndarray, qd, w: np.ndarray, s_ref: float, cs=None, sc: float = 0.0, abs_mode: bool = False):
    """One query: gated values qg [D], ungated values qd [C] or None -> dict(q8, u, shift, u_f, q8u, sq)."""
    qop = np.abs(qg.astype(np.float64)) if abs_mode else np.maximum(qg.astype(np.float64), 0.0)
    qw = qop * w
    gm = float(qw.max()) if qw.size else 0.0
    u_nat = max(gm * (1.00001 / 255.0) * s_ref, 1e-30)
    shift = 0
    q8u = None
    sq = 1.0
    if qd is not None and qd.size and sc > 0 and float(np.abs(qd.astype(np.float64) * (cs / sc)).max()) > 0:
        qp = qd.astype(np.float64) * (cs / sc)
        v_nat = sc * float(np.abs(qp).max()) / 127.0
        if u_nat >= v_nat:
            shift = int(min(max_shift(qg.shape[0]), max(0, np.floor(np.log2(u_nat / v_nat)))))
            u_f = u_nat / 2.0 ** shift
        else:
            u_f = v_nat
        sq = u_f / sc
        q8u = I8.quant(qp, sq)
    else:
        u_f = u_nat
        if qd is not None and qd.size and sc > 0:
            q8u = np.zeros(qd.shape[0])
            sq = u_f / sc
    u = u_f * 2.0 ** shift
    q8 = np.minimum(np.ceil(qw / (u / s_ref) * (1.0 + 1e-6)), 255.0)
    return dict(q8=q8, u=u, shift=shift, u_f=u_f, q8u=q8u, sq=sq)


def bound_scores(qg, qi, qd, cg, ci, cd, bucket_q=None, bucket_d=None, abs_mode=False):
    """Bound scores [Q, N] of a gated_i8 index in float64 (nominal arithmetic) and the per-query margins.
    bucket_q / bucket_d: bucket of every (query, slice) / (row, slice); default: buckets == index values (then a bucket match IS an
    index match and the bound differs from the exact score only by the roundings)."""
    if bucket_q is None:
        bucket_q, bucket_d = qi, ci
    step, w, s_ref = corpus_steps(cg)
    d8 = corpus_image(cg, step)
    has_u = cd is not None and cd.shape[1] > 0
    if has_u:
        d8u, cs, sc, ec, nc = I8.corpus_image(cd)
    U = np.empty((qg.shape[0], cg.shape[0]))
    margin = np.zeros(qg.shape[0])
    for q in range(qg.shape[0]):
        r = query_units(qg[q], qd[q] if has_u else None, w, s_ref, cs if has_u else None, sc if has_u else 0.0, abs_mode)
        m = bucket_d == bucket_q[q][None, :]
        U[q] = r["u"] * ((d8 * m) @ r["q8"])
        if has_u and r["q8u"] is not None:
            U[q] += r["u_f"] * (d8u @ r["q8u"])
            qp = qd[q].astype(np.float64) * (cs / sc)
            margin[q] = np.linalg.norm(qp) * ec + np.linalg.norm(qp - r["sq"] * r["q8u"]) * nc
    return U, margin
