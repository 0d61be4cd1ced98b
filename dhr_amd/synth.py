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


# ----------------------------------------------------------------------------- structured dense columns ("--data clustered")
# SURVEY.md 8(d) specifies i.i.d. Gaussian dense columns; real encoder output is not: it clusters, its spectrum decays, corpora hold
# near-duplicate passages, and some queries sit on a cluster centre.  The filter of this design is data-dependent, so the
# bench can also be run on this variant: N_CLUSTERS Gaussian clusters with a skewed size distribution (p_i ~ (1 + i)^-0.7), sigma_within =
# 0.3 sigma_between, per-dimension standard deviation ~ (1 + d / 32)^-1/2 (the overall scale keeps the mean per-column std at 0.1),
# DUP_FRAC of the rows near-duplicates of another row (the WHOLE row: gated values, indices, dense part + 1 % noise), HOT_FRAC of the queries
# on a cluster centre (+ 5 % noise).
N_CLUSTERS = 2000
SIGMA_WITHIN = 0.3
DUP_FRAC = 0.01
HOT_FRAC = 0.05


def cluster_model(seed: int, d: int):
    """-> (centres [N_CLUSTERS, d] fp32 in units of the final scale, per-dimension std of the within-cluster noise [d], cluster cdf)."""
    rng = np.random.Generator(np.random.PCG64(seed * 7919 + 17))
    spec = (1.0 + np.arange(d) / 32.0) ** -0.5
    scale = 0.1 / np.sqrt((1.0 + SIGMA_WITHIN ** 2) * np.mean(spec ** 2))
    centres = (rng.standard_normal((N_CLUSTERS, d)) * spec * scale).astype(np.float32)
    p = (1.0 + np.arange(N_CLUSTERS)) ** -0.7
    return centres, (SIGMA_WITHIN * spec * scale).astype(np.float32), np.cumsum(p / p.sum())


def make_dense_clustered(rng: np.random.Generator, n: int, d: int, model, *, hot_frac: float = 0.0) -> np.ndarray:
    centres, sw, cdf = model
    c = np.minimum(np.searchsorted(cdf, rng.random(n)), N_CLUSTERS - 1)
    noise = rng.standard_normal((n, d)).astype(np.float32) * sw
    hot = rng.random(n) < hot_frac
    noise[hot] *= 0.05 / SIGMA_WITHIN
    return (centres[c] + noise).astype(np.float16)


def torch_make_dense_clustered(gen, n: int, d: int, model_t, device, *, hot_frac: float = 0.0):
    """torch form on `device`; model_t = (centres, sw, cdf) as device tensors (torch_cluster_model)."""
    import torch
    centres, sw, cdf = model_t
    c = torch.bucketize(torch.rand((n,), generator=gen, device=device), cdf).clamp_(max=N_CLUSTERS - 1)
    noise = torch.randn((n, d), generator=gen, device=device) * sw
    if hot_frac > 0:
        hot = torch.rand((n, 1), generator=gen, device=device) < hot_frac
        noise = torch.where(hot, noise * (0.05 / SIGMA_WITHIN), noise)
    return (centres[c] + noise).to(torch.float16)


def torch_cluster_model(seed: int, d: int, device):
    import torch
    centres, sw, cdf = cluster_model(seed, d)
    return (torch.from_numpy(centres).to(device), torch.from_numpy(sw).to(device), torch.tensor(cdf, device=device, dtype=torch.float32))


def torch_near_duplicates(gen, value, index, d_dlr: int, frac: float = DUP_FRAC):
    """In place: a fraction `frac` of the rows become copies of another row of the block (gated values and indices exactly, dense part
    times (1 + 1 % noise))."""
    import torch
    n = value.shape[0]
    m = int(n * frac)
    if m <= 0:
        return
    dst = torch.randint(0, n, (m,), generator=gen, device=value.device)
    src = torch.randint(0, n, (m,), generator=gen, device=value.device)
    value[dst] = value[src]
    if index is not None:
        index[dst] = index[src]
    if value.shape[1] > d_dlr:
        jit = 1.0 + 0.01 * torch.randn((m, value.shape[1] - d_dlr), generator=gen, device=value.device)
        value[dst, d_dlr:] = (value[dst, d_dlr:].float() * jit).to(torch.float16)
