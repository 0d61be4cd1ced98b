"""Shared helpers for the test-suite: fixture loading and the parity rule."""
import json
import os
import pickle
from types import SimpleNamespace

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self):
        self.calls = dict(np.load(os.path.join(GOLDEN_DIR, "golden_calls.npz"), allow_pickle=False))
        with open(os.path.join(GOLDEN_DIR, "golden_meta.json")) as f:
            self.meta = json.load(f)
        self._inputs = {}

    def inputs(self, name):
        if name not in self._inputs:
            z = np.load(os.path.join(GOLDEN_DIR, f"inputs_{name}.npz"), allow_pickle=False)
            d = {k: z[k] for k in z.files}
            d.setdefault("ci", None)
            d.setdefault("qi", None)
            self._inputs[name] = d
        return self._inputs[name]

    def case(self, name):
        info = dict(self.meta["cases"][name])
        return info, self.calls[name + ".rows"], self.calls[name + ".scores"]

    def trec(self, name):
        with open(os.path.join(GOLDEN_DIR, name), "rb") as f:
            return f.read().decode()


def case_args(info):
    return SimpleNamespace(emb_dim=info.get("emb_dim", 768), theta=info.get("theta", 0.1),
                           topk=info["topk"], agip_topk=info.get("agip_topk", 10000),
                           IP=info.get("IP", False), brute_force=info.get("brute_force", False),
                           rerank=info.get("rerank", False))


def dump_pickle(path, value, index, ids):
    with open(path, "wb") as f:
        pickle.dump([value, index, list(ids)], f, protocol=4)


def parse_trec(text):
    """-> dict qid -> list of (docid, rank, score)"""
    out = {}
    for line in text.splitlines():
        qid, _, docid, rank, score, _run = line.split(" ")
        out.setdefault(qid, []).append((docid, int(rank), float(score)))
    return out


def sorted_lists(rng, n_lists, q, ll, ragged=True):
    """[n_lists, q, ll] per-shard result lists as the search writes them: (score desc, row asc), (-inf, -1) tails."""
    s = np.round(rng.standard_normal((n_lists, q, ll)).astype(np.float32), 1) + np.float32(0)   # many exact ties; no -0.0
    r = rng.permutation(10_000_000)[: n_lists * q * ll].reshape(n_lists, q, ll).astype(np.int64)
    for l in range(n_lists):
        for i in range(q):
            order = np.lexsort((r[l, i], -s[l, i].astype(np.float64)))
            s[l, i], r[l, i] = s[l, i][order], r[l, i][order]
            if ragged:
                fill = int(rng.integers(0, ll + 1))
                s[l, i, fill:], r[l, i, fill:] = -np.inf, -1
    return s, r
