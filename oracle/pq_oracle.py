"""ORACLE (test infrastructure, NOT product code): numpy restatement of the product quantiser behind `--PQIP`.

The reference calls faiss `IndexPQ(d, M, nbits=8, METRIC_INNER_PRODUCT)` (retrieval/quantize_index.py:27-37,
retrieval/gip_retrieval.py:170,202); faiss is a third-party dependency that is NOT in /root/reference (no version is pinned
there either), so this file restates its published algorithm (Jegou et al., "Product quantization for nearest neighbor
search"; faiss ProductQuantizer / IndexPQ): per-subspace k-means with 256 centroids, nearest-centroid codes (L2), asymmetric
distance computation for the inner product: score(q, x) = sum_m <q_m, c_m[code_m(x)]>.

PARITY UNPINNED: there are no golden vectors for this path and faiss cannot be run here.  Training is only checked for
quality (the product's codebooks must quantise as well as this restatement's), encoding / decoding / ADC search are checked
exactly against this file with the SAME codebooks.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np


def train(x: np.ndarray, M: int, iters: int = 25, max_points: int = 65536, nbits: int = 8):
    """Lloyd k-means per subspace, the same deterministic schedule as dhr_pq_train (every stride-th row, the first 256
    evenly spaced training rows as initial centroids, empty clusters re-seeded from the largest one)."""
    n, d = x.shape
    dsub = d // M
    stride = max(1, n // max_points)
    t = x[::stride].astype(np.float32)
    npnts = t.shape[0]
    ksub = 1 << nbits                                  # faiss.IndexPQ(d, M, nbits): 2^nbits centroids per sub-quantiser
    cb = np.empty((M, ksub, dsub), np.float32)
    pick = (np.arange(ksub) * (npnts // ksub)) if npnts >= ksub else (np.arange(ksub) % npnts)
    for m in range(M):
        sub = t[:, m * dsub:(m + 1) * dsub]
        c = sub[pick].copy()
        for _ in range(iters):
            a = assign(sub, c)
            cnt = np.bincount(a, minlength=ksub)
            sums = np.zeros_like(c)
            np.add.at(sums, a, sub)
            big = int(np.argmax(cnt))
            for k in range(ksub):
                if cnt[k] > 0:
                    c[k] = sums[k] / cnt[k]
                else:
                    sign = np.where(((np.arange(dsub) + k) & 1) == 1, 1.0, -1.0).astype(np.float32)
                    c[k] = sums[big] / max(cnt[big], 1) * (1.0 + sign / 1024.0)
        cb[m] = c
    return cb


def assign(sub: np.ndarray, c: np.ndarray) -> np.ndarray:
    """nearest centroid (L2), first minimum: argmin_k |c_k|^2 - 2 <x, c_k>."""
    dist = (c * c).sum(1)[None, :] - 2.0 * sub.astype(np.float32) @ c.T
    return dist.argmin(1)


def encode(x: np.ndarray, cb: np.ndarray) -> np.ndarray:
    M, _, dsub = cb.shape
    return np.stack([assign(x[:, m * dsub:(m + 1) * dsub], cb[m]) for m in range(M)], 1).astype(np.uint8)


def decode(codes: np.ndarray, cb: np.ndarray) -> np.ndarray:
    M = cb.shape[0]
    return np.concatenate([cb[m][codes[:, m]] for m in range(M)], 1)


def mse(x: np.ndarray, cb: np.ndarray) -> float:
    r = decode(encode(x, cb), cb) - x.astype(np.float32)
    return float((r * r).sum(1).mean())


def adc_scores(q: np.ndarray, codes: np.ndarray, cb: np.ndarray) -> np.ndarray:
    """[Q, N] asymmetric inner-product scores from per-query lookup tables (what IndexPQ.search ranks by)."""
    M, _, dsub = cb.shape
    out = np.zeros((q.shape[0], codes.shape[0]), np.float64)
    for m in range(M):
        table = q[:, m * dsub:(m + 1) * dsub].astype(np.float64) @ cb[m].T.astype(np.float64)      # [Q, 256]
        out += table[:, codes[:, m]]
    return out
