"""ORACLE (test infrastructure, NOT product code).

CPU restatement, in numpy, of castorini/dhr's brute-force dense-hybrid retrieval path
(/root/reference/retrieval/gip_retrieval.py + retrieval/index.py + retrieval/merge.result.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (dhr_amd/) never does.

Parity pin: every function here is checked against outputs of the reference itself, run in the
build container by tests/golden/make_golden.py (the reference is imported from /root/reference with
`pickle5`/`faiss`/`progressbar` stubbed) and committed as fixtures under tests/golden/*.npz|*.trec.
tests/test_oracle_golden.py replays them.  The PQ first stage (`PQ_IP_retrieval`, faiss, absent
here) is restated separately in oracle/pq_oracle.py from the published algorithm: parity unpinned for that row.

All file:line citations are relative to /root/reference/.
"""
from __future__ import annotations

import glob
import os
import pickle
from collections import defaultdict
from types import SimpleNamespace

import numpy as np

NEG_INF = float("-inf")


# --------------------------------------------------------------------------- scoring kernels (CPU)
def pad_idx(idx: np.ndarray, cls_dim: int) -> np.ndarray:
    """F.pad(idx, (0, cls_dim), value=1) -- retrieval/gip_retrieval.py:110-113."""
    if cls_dim <= 0:
        return idx
    return np.pad(idx, ((0, 0), (0, cls_dim)), mode="constant", constant_values=1)


def gip_scores_f32(q_val: np.ndarray, q_idx: np.ndarray, c_val: np.ndarray, c_idx: np.ndarray) -> np.ndarray:
    """One query against the corpus, the reference's op sequence in fp32
    (gip_retrieval.py:119-120): mask = (c_idx == q_idx); tmp = mask * c_val; einsum('ij,j->i').
    q_idx / c_idx are already padded over the CLS tail."""
    tmp = (c_idx == q_idx[None, :]) * c_val
    return np.einsum("ij,j->i", tmp, q_val, optimize=False).astype(np.float32, copy=False)


def ip_scores_f32(q_val: np.ndarray, c_val: np.ndarray) -> np.ndarray:
    """einsum('ij,j->i', corpus, query) -- gip_retrieval.py:74."""
    return (c_val @ q_val).astype(np.float32, copy=False)


def gip_scores_f64(q_val: np.ndarray, q_idx, c_val: np.ndarray, c_idx) -> np.ndarray:
    """Exact (float64) gated inner product; the tie/epsilon rule of the parity tests is decided on
    these.  q_val is the fp32 query the reference actually uses (after the lamda scaling)."""
    qv = q_val.astype(np.float64)
    cv = c_val.astype(np.float64)
    if q_idx is None or c_idx is None:
        return cv @ qv
    d = q_idx.shape[-1]
    gate = (c_idx[:, :d] == q_idx[None, :d])
    s = (gate * cv[:, :d]) @ qv[:d]
    if cv.shape[1] > d:
        s = s + cv[:, d:] @ qv[d:]
    return s


def topk_desc(scores: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest scores, best first.  torch.topk / argsort tie order is
    implementation-defined in the reference (SURVEY.md section 7 hard part 2); the oracle breaks
    ties by lower row index, and the tests compare tie groups as sets."""
    k = min(k, scores.shape[0])
    order = np.lexsort((np.arange(scores.shape[0]), -scores.astype(np.float64)))
    return order[:k]


# --------------------------------------------------------------------------- reference functions
def IP_retrieval(qids, query_embs, corpus_embs, args):
    """gip_retrieval.py:60-85.  argsort(desc)[:topk]: silently returns N rows when topk > N."""
    all_results, all_scores = {}, {}
    for i, q in enumerate(query_embs):
        s = ip_scores_f32(q, corpus_embs)
        cand = topk_desc(s, args.topk)
        all_scores[qids[i]] = s[cand].tolist()
        all_results[qids[i]] = cand.tolist()
    return all_results, all_scores


def GIP_retrieval(qids, query_embs, query_arg_idxs, corpus_embs, corpus_arg_idxs, args):
    """gip_retrieval.py:88-165 (brute force, theta>0 one-stage, and the two-stage --rerank modes)."""
    theta = 0 if args.brute_force else args.theta                      # :89-91
    cls_dim = query_embs.shape[1] - args.emb_dim                       # :110
    q_idx_p = pad_idx(query_arg_idxs, cls_dim)
    c_idx_p = pad_idx(corpus_arg_idxs, cls_dim)
    n = corpus_embs.shape[0]
    all_results, all_scores = {}, {}
    for i, (q, qi) in enumerate(zip(query_embs, q_idx_p)):
        if theta == 0:                                                 # :117-126
            if args.topk > n:
                raise RuntimeError("selected index k out of range")   # torch.topk behaviour
            s = gip_scores_f32(q, qi, corpus_embs, c_idx_p)
            cand = topk_desc(s, args.topk)
            sc = s[cand]
        else:                                                          # :128-156
            important = np.nonzero(q > theta)[0]                       # topk(q, num_idx) == this set
            if not args.IP:
                partial = gip_scores_f32(q[important], qi[important], corpus_embs[:, important],
                                         c_idx_p[:, important])        # :135-136
            else:
                partial = ip_scores_f32(q, corpus_embs)                # :139
            if args.rerank:
                if args.agip_topk > n:
                    raise RuntimeError("selected index k out of range")
                c1 = topk_desc(partial, args.agip_topk)                # :142
                s2 = gip_scores_f32(q, qi, corpus_embs[c1], c_idx_p[c1])   # :144-146
                o = topk_desc(s2, args.topk)                           # :148
                cand, sc = c1[o], s2[o]
            else:
                if args.topk > n:
                    raise RuntimeError("selected index k out of range")
                cand = topk_desc(partial, args.topk)                   # :155
                sc = partial[cand]
        all_scores[qids[i]] = sc.tolist()
        all_results[qids[i]] = cand.tolist()
    return all_results, all_scores


# --------------------------------------------------------------------------- main(): load / shard / write
def prepare_queries(query_embs, query_arg_idxs, emb_dim: int, lamda: float):
    """gip_retrieval.py:268-283 (CPU branch): fp16 -> fp32, idx None stays None, CLS tail *= lamda
    (in fp32, scalar cast to fp32 as torch does)."""
    q = np.asarray(query_embs).astype(np.float32)
    qi = None if (query_arg_idxs is None or np.isscalar(query_arg_idxs)) else np.asarray(query_arg_idxs)
    cls_dim = q.shape[1] - emb_dim
    if cls_dim > 0:
        q[:, -cls_dim:] = np.float32(lamda) * q[:, -cls_dim:]
    return q, qi


def shard_rows(n_docs: int, total_shrad: int, shrad: int):
    """Row range of one shard -- gip_retrieval.py:292-306: per = len(docids)//total; the last shard
    runs to the end."""
    per = n_docs // total_shrad
    lo = per * shrad
    hi = n_docs if shrad == total_shrad - 1 else per * (shrad + 1)
    return lo, hi


def prepare_corpus(corpus_embs, corpus_arg_idxs, docids, total_shrad: int = 1, shrad: int = 0):
    """gip_retrieval.py:289-315 (CPU branch).  A merged dense index stores the int 0 as its index
    array (index.py:40-43); slicing it raises and maps it to None (:295-305)."""
    lo, hi = shard_rows(len(docids), total_shrad, shrad)
    c = np.asarray(corpus_embs)[lo:hi].astype(np.float32)
    if corpus_arg_idxs is None or np.isscalar(corpus_arg_idxs):
        ci = None
    else:
        ci = np.asarray(corpus_arg_idxs)[lo:hi]
    return c, ci, list(docids[lo:hi])


def trec_lines(results, scores, docids, run_name: str = "h2oloo"):
    """gip_retrieval.py:333-342: '{qid} Q0 {docid} {rank+1} {score} {run_name}', skipping
    docid == query_id, rank numbers keep their gaps."""
    out = []
    for query_id in results:
        result, score = results[query_id], scores[query_id]
        for rank, docidx in enumerate(result):
            docid = docids[docidx]
            if docid != query_id:
                out.append("{} Q0 {} {} {} {}\n".format(query_id, docid, rank + 1, score[rank], run_name))
    return out


def run_main(query_pickle: str, index_pickle: str, *, emb_dim=768, theta=0.1, topk=1000, agip_topk=10000,
             IP=False, brute_force=False, rerank=False, lamda=1.0, total_shrad=1, shrad=0,
             run_name="h2oloo"):
    """The whole of main() (gip_retrieval.py:233-344) minus argparse; returns the TREC text."""
    args = SimpleNamespace(emb_dim=emb_dim, theta=theta, topk=topk, agip_topk=agip_topk, IP=IP,
                           brute_force=brute_force, rerank=rerank)
    with open(query_pickle, "rb") as f:
        query_embs, query_arg_idxs, qids = pickle.load(f)
    q, qi = prepare_queries(query_embs, query_arg_idxs, emb_dim, lamda)
    with open(index_pickle, "rb") as f:
        corpus_embs, corpus_arg_idxs, docids = pickle.load(f)
    c, ci, docids = prepare_corpus(corpus_embs, corpus_arg_idxs, docids, total_shrad, shrad)
    if qi is not None:
        results, scores = GIP_retrieval(qids, q, qi, c, ci, args)
    else:
        results, scores = IP_retrieval(qids, q, c, args)
    return "".join(trec_lines(results, scores, docids, run_name))


# --------------------------------------------------------------------------- index.py / merge.result.py
def merge_index(index_path: str, index_prefix: str, order=None):
    """retrieval/index.py:26-47.  The reference iterates glob order (unsorted, filesystem dependent);
    `order` (list of basenames) replays a recorded order, default is sorted."""
    if order is not None:
        files = [os.path.join(index_path, b) for b in order]
    else:
        files = sorted(glob.glob(os.path.join(index_path, index_prefix + ".split*.pt")))
    embs, idxs, docids = [], [], []
    for fn in files:
        with open(fn, "rb") as f:
            e, i, d = pickle.load(f)
        embs.append(e)
        idxs.append(i)
        docids += d
    try:
        idxs = np.concatenate(idxs, axis=0)
    except Exception:
        idxs = 0                                                       # index.py:40-43
    return [np.concatenate(embs, axis=0), idxs, docids]


def merge_results(shard_texts, topk: int = 1000, run_name: str = "dhr"):
    """retrieval/merge.result.py:22-42 on in-memory TREC texts (one per shard, in shard order)."""
    results, scores = defaultdict(list), defaultdict(list)
    for text in shard_texts:
        for line in text.splitlines():
            query_id, _, docid, _rank, score, _ = line.strip().split(" ")
            results[query_id].append(docid)
            scores[query_id].append(float(score))
    out = []
    for query_id in results:
        score, result = scores[query_id], results[query_id]
        sort_idx = np.array(score).argsort()[::-1][:topk]
        for rank, idx in enumerate(sort_idx):
            out.append("{} Q0 {} {} {} {}\n".format(query_id, result[idx], rank + 1, score[idx], run_name))
    return "".join(out)


def merge_topk(shard_scores, shard_rows_, k: int):
    """Array form of the same reduce: per query, concat the shards' (score, global row) lists and
    keep the k best (score desc, row asc).  shard_scores/rows: list of [Q, k_s] arrays; unfilled
    slots are (-inf, -1)."""
    s = np.concatenate(shard_scores, axis=1)
    r = np.concatenate(shard_rows_, axis=1)
    q = s.shape[0]
    out_s = np.full((q, k), NEG_INF, np.float32)
    out_r = np.full((q, k), -1, np.int64)
    for i in range(q):
        valid = r[i] >= 0
        si, ri = s[i][valid], r[i][valid]
        order = np.lexsort((ri, -si.astype(np.float64)))[:k]
        out_s[i, : len(order)] = si[order]
        out_r[i, : len(order)] = ri[order]
    return out_s, out_r


# --------------------------------------------------------------------------- parity rule used by the tests
def check_topk(rows, scores, exact_f64: np.ndarray, k: int, *, eps_rel: float = 1e-5, atol: float = 1e-3,
               ref_scores=None):
    """Contract of BASELINE.md section 3: the returned set equals the exact top-k except for docs
    whose float64 score lies within eps of the k-th best; scores within atol.  Returns a dict of
    diagnostics, raises AssertionError on violation."""
    rows = np.asarray(rows, np.int64)
    scores = np.asarray(scores, np.float64)
    n = exact_f64.shape[0]
    kk = min(k, n)
    assert len(rows) == kk, f"expected {kk} rows, got {len(rows)}"
    assert len(set(rows.tolist())) == kk, "duplicate rows in the result"
    assert np.all((rows >= 0) & (rows < n))
    assert np.all(np.abs(scores - exact_f64[rows]) <= atol), \
        f"score error {np.abs(scores - exact_f64[rows]).max()}"
    assert np.all(np.diff(scores) <= 1e-6 * np.maximum(1.0, np.abs(scores[:-1]))), "not sorted best-first"
    if kk == n:
        return {"boundary": 0}
    sk = np.sort(exact_f64)[::-1][kk - 1]
    eps = eps_rel * max(abs(sk), 1e-30) + 1e-12
    must = set(np.nonzero(exact_f64 > sk + eps)[0].tolist())
    may = set(np.nonzero(exact_f64 >= sk - eps)[0].tolist())
    got = set(rows.tolist())
    assert must <= got, f"missing {len(must - got)} rows strictly above the boundary"
    assert got <= may, f"{len(got - may)} returned rows are strictly below the boundary"
    if ref_scores is not None:
        assert np.all(np.abs(np.sort(scores) - np.sort(np.asarray(ref_scores, np.float64))) <= atol)
    return {"boundary": len(may) - len(must)}


# --------------------------------------------------------------------------- two-stage parity rule
def stage1_scores_f64(q_val, q_idx, c_val, c_idx, theta: float, ip: bool) -> np.ndarray:
    """Float64 stage-1 score of the theta>0 modes (gip_retrieval.py:128-139): the gated inner product restricted to the
    query dimensions with q > theta, or the plain inner product over all columns with --IP.  q_idx / c_idx unpadded."""
    qv = np.asarray(q_val, np.float64)
    cv = np.asarray(c_val, np.float64)
    if ip:
        return cv @ qv
    d = q_idx.shape[-1]
    keep = qv > theta
    gate = (c_idx[:, :d] == q_idx[None, :d]) & keep[None, :d]
    s = (gate * cv[:, :d]) @ qv[:d]
    if cv.shape[1] > d:
        s = s + cv[:, d:] @ (qv[d:] * keep[d:])
    return s


def check_two_stage(rows, scores, stage1_f64: np.ndarray, exact_f64: np.ndarray, k1: int, k: int, *,
                    eps_rel: float = 1e-5, atol: float = 1e-3):
    """Parity rule of `--rerank` (gip_retrieval.py:141-153): the result is the exact-score top-k of SOME stage-1 set C1
    with |C1| = k1 that holds every row strictly above the k1-th best stage-1 score and nothing strictly below it.  A
    returned list may differ from the reference's only by rows inside the float64 tie band of the stage-1 boundary
    (1e-5 relative at the k1-th stage-1 score) or of the stage-2 boundary (1e-5 relative at the k-th exact score)."""
    rows = np.asarray(rows, np.int64)
    scores = np.asarray(scores, np.float64)
    n = exact_f64.shape[0]
    k1 = min(k1, n)
    kk = min(k, k1)
    assert len(rows) == kk and len(set(rows.tolist())) == kk
    assert np.all(np.abs(scores - exact_f64[rows]) <= atol)
    assert np.all(np.diff(scores) <= 1e-6 * np.maximum(1.0, np.abs(scores[:-1]))), "not sorted best-first"
    s1k = np.sort(stage1_f64)[::-1][k1 - 1]
    e1 = eps_rel * max(abs(s1k), 1e-30) + 1e-12
    must1 = stage1_f64 > s1k + e1
    may1 = stage1_f64 >= s1k - e1
    assert np.all(may1[rows]), "a returned row is strictly below the stage-1 boundary"
    # stage 2: k-th returned exact score; every CERTAIN stage-1 row strictly above it must be present
    sk = exact_f64[rows].min()
    e2 = eps_rel * max(abs(sk), 1e-30) + 1e-12
    got = np.zeros(n, bool)
    got[rows] = True
    missing = must1 & ~got & (exact_f64 > sk + e2)
    assert not missing.any(), f"{int(missing.sum())} certain stage-1 rows beat the k-th returned score but are missing"
    # ... and the list cannot be better than any admissible C1 allows: with C1 >= must1 the k-th best of C1 is at
    # least the k-th best exact score inside must1
    if int(must1.sum()) >= kk:
        lower = np.sort(exact_f64[must1])[::-1][kk - 1]
        assert sk >= lower - e2
    return {"band1": int(may1.sum() - must1.sum())}
