"""Synthetic [densified-lexical || dense] vectors in the reference's on-disk format.

The reference ships no data generator; this follows SURVEY.md section 8(d): DLR columns are
produced *through the densify rule* (vocab id t -> slice t % D, index t // D, keep the max on
collision: /root/reference/tevatron/DHR/utils.py:5-22, densify/densify_corpus.py:29-52) so the
index statistics look like an encoder's, dense columns are fp16(N(0,1)*0.1).  Everything is
rounded to fp16 (the file format, tevatron/driver/encode.py:165-170) before use.

Two implementations with the same distribution: numpy (CPU tests, fixtures) and torch (bench, on
the GPU).  They are not bit-identical to each other and do not need to be.
"""
from __future__ import annotations

import numpy as np

VOCAB = 30522 - 570          # wordpiece vocab minus the unused tokens (DHR/utils.py:8)
BG_MAX = 0.02                # "softmax-leak" background magnitude


def _zipf_p(v: int, a: float = 1.1) -> np.ndarray:
    p = np.arange(1, v + 1, dtype=np.float64) ** -a
    return p / p.sum()


def bg_categorical(n_idx: int = 39) -> np.ndarray:
    """Fixed skewed categorical over the index values used by background slices."""
    p = np.arange(1, n_idx + 1, dtype=np.float64) ** -1.0
    return p / p.sum()


def make_dlr(rng: np.random.Generator, n: int, d_dlr: int, lmin: int, lmax: int, *,
             vocab: int = VOCAB, background: bool = True, uniform_idx: bool = False,
             idx_dtype=np.uint8, integer_weights: bool = False, perm: np.ndarray | None = None):
    """Return (value fp16 [n,d_dlr], index idx_dtype [n,d_dlr])."""
    n_idx = (vocab + d_dlr - 1) // d_dlr
    val = np.zeros((n, d_dlr), np.float32)
    idx = np.zeros((n, d_dlr), np.int64)
    if background:
        val[:] = rng.uniform(0.0, BG_MAX, (n, d_dlr)).astype(np.float32)
        if uniform_idx:
            idx[:] = rng.integers(0, n_idx, (n, d_dlr))
        else:
            idx[:] = rng.choice(n_idx, size=(n, d_dlr), p=bg_categorical(n_idx))
    if perm is None:
        perm = np.random.Generator(np.random.PCG64(977)).permutation(vocab)
    length = rng.integers(lmin, lmax + 1, n)
    t = perm[rng.choice(vocab, size=(n, lmax), p=_zipf_p(vocab))]
    if integer_weights:
        w = rng.integers(1, 4, (n, lmax)).astype(np.float32)
    else:
        w = rng.uniform(0.1, 3.0, (n, lmax)).astype(np.float32)
    live = np.arange(lmax)[None, :] < length[:, None]
    rows = np.broadcast_to(np.arange(n)[:, None], (n, lmax))[live]
    tt, ww = t[live], w[live]
    order = np.argsort(ww, kind="stable")          # ascending: the largest weight is assigned last
    flat = rows[order] * d_dlr + tt[order] % d_dlr
    val.reshape(-1)[flat] = ww[order]              # numpy: on repeated indices the last one wins
    idx.reshape(-1)[flat] = tt[order] // d_dlr
    return val.astype(np.float16), idx.astype(idx_dtype)


def make_dense(rng: np.random.Generator, n: int, d: int) -> np.ndarray:
    return (rng.standard_normal((n, d)) * 0.1).astype(np.float16)


def make_pair(seed: int, n: int, q: int, d_dlr: int = 768, d_cls: int = 768, *, kind: str = "encoder",
              uniform_idx: bool = False):
    """(corpus_value, corpus_index|None, query_value, query_index|None) for one config.

    kind: "encoder" (uint8 idx, background) | "bm25" (int16 idx, sparse, whole-word vocab) |
          "dense" (no index array)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "dense":
        return make_dense(rng, n, d_cls), None, make_dense(rng, q, d_cls), None
    if kind == "bm25":
        vocab = 2_600_000
        perm = None if vocab == VOCAB else np.arange(vocab)
        cv, ci = make_dlr(rng, n, d_dlr, 30, 90, vocab=vocab, background=False, idx_dtype=np.int16, perm=perm)
        qv, qi = make_dlr(rng, q, d_dlr, 4, 12, vocab=vocab, background=False, idx_dtype=np.int16,
                          integer_weights=True, perm=perm)
    else:
        cv, ci = make_dlr(rng, n, d_dlr, 30, 90, uniform_idx=uniform_idx)
        qv, qi = make_dlr(rng, q, d_dlr, 4, 12, uniform_idx=uniform_idx)
    if d_cls > 0:
        cv = np.concatenate([cv, make_dense(rng, n, d_cls)], axis=1)
        qv = np.concatenate([qv, make_dense(rng, q, d_cls)], axis=1)
    return cv, ci, qv, qi


# ----------------------------------------------------------------------------- torch (bench, on device)
def torch_make_dlr(gen, n: int, d_dlr: int, lmin: int, lmax: int, device, *, uniform_idx: bool = False,
                   vocab: int = VOCAB):
    """Same distribution as make_dlr(background=True) with torch ops on `device`.
    Returns (value fp16 [n,d_dlr], index uint8 [n,d_dlr])."""
    import torch
    n_idx = (vocab + d_dlr - 1) // d_dlr
    val = torch.rand((n, d_dlr), generator=gen, device=device, dtype=torch.float32) * BG_MAX
    if uniform_idx:
        idx = torch.randint(0, n_idx, (n, d_dlr), generator=gen, device=device, dtype=torch.int64)
    else:
        cdf = torch.tensor(np.cumsum(bg_categorical(n_idx)), device=device, dtype=torch.float32)
        u = torch.rand((n, d_dlr), generator=gen, device=device)
        idx = torch.bucketize(u, cdf).clamp_(max=n_idx - 1)
    perm = torch.from_numpy(np.random.Generator(np.random.PCG64(977)).permutation(vocab)).to(device)
    cz = torch.tensor(np.cumsum(_zipf_p(vocab)), device=device, dtype=torch.float32)
    t = perm[torch.bucketize(torch.rand((n, lmax), generator=gen, device=device), cz).clamp_(max=vocab - 1)]
    w = torch.rand((n, lmax), generator=gen, device=device) * 2.9 + 0.1
    length = torch.randint(lmin, lmax + 1, (n, 1), generator=gen, device=device)
    w = torch.where(torch.arange(lmax, device=device)[None, :] < length, w, torch.zeros_like(w))
    live = w > 0
    dummy = torch.full_like(t, d_dlr)                      # dead / losing entries land in a spare column
    sl = torch.where(live, t % d_dlr, dummy)
    hv = torch.zeros((n, d_dlr + 1), device=device)
    hv.scatter_reduce_(1, sl, w, reduce="amax", include_self=True)   # keep the max on collision
    won = live & (w == hv.gather(1, sl))
    hi = torch.full((n, d_dlr + 1), -1, device=device, dtype=torch.int64)
    hi.scatter_(1, torch.where(won, sl, dummy), t // d_dlr)
    hv, hi = hv[:, :d_dlr], hi[:, :d_dlr]
    has = hi >= 0
    val = torch.where(has, hv, val)
    idx = torch.where(has, hi, idx)
    return val.to(torch.float16), idx.to(torch.uint8)
