#!/usr/bin/env python
"""Degenerate inputs through the C ABI against the oracle, both images of the gated half: an all-zero corpus / all-zero queries / identical rows (every
score ties: the lists are "row ascending"), k == n, one row, fp16 maxima and subnormals, all-negative values on either side, one non-zero column.
usage: python tools/degenerate.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def run(name, cv, ci, qv, qi, k):
    from dhr_amd.retrieval import gip_retrieval as G
    from oracle import gip_oracle as O
    for g8 in (0, 1):
        os.environ["DHR_GATED_I8"] = str(g8)
        ix = G.GipIndex(cv, ci)
        try:
            s, r = ix.search(qv, qi, k)
        finally:
            ix.close()
        c32 = cv.astype(np.float32)
        for i in range(qv.shape[0]):
            ex = O.gip_scores_f64(qv[i].astype(np.float32), None if qi is None else qi[i], c32, ci)
            O.check_topk(r[i], s[i], ex, k)
            if np.all(ex == ex[0]):                        # every score ties: the order is "row ascending"
                assert r[i].tolist() == list(range(min(k, len(ex)))), (name, r[i][:10])
    print(name, "ok", flush=True)


def main():
    rng = np.random.default_rng(0)
    for n, d_dlr, d_cls in ((5000, 64, 32), (70000, 128, 0), (300, 0, 64), (1, 8, 8), (257, 768, 768)):
        K = d_dlr + d_cls
        q = 5
        ci = rng.integers(0, 5, (n, d_dlr)).astype(np.uint8) if d_dlr else None
        qi = rng.integers(0, 5, (q, d_dlr)).astype(np.uint8) if d_dlr else None
        cvr = (np.abs(rng.standard_normal((n, K))) * 0.3).astype(np.float16)
        qvr = (np.abs(rng.standard_normal((q, K))) * 0.3).astype(np.float32)
        k = min(n, 100)
        run("zero corpus %d" % n, np.zeros((n, K), np.float16), ci, qvr, qi, k)
        run("zero queries %d" % n, cvr, ci, np.zeros((q, K), np.float32), qi, k)
        run("identical rows %d" % n, np.repeat(cvr[:1], n, 0), None if ci is None else np.repeat(ci[:1], n, 0), qvr, qi, k)
        run("k == n %d" % n, cvr, ci, qvr, qi, min(n, 16384))
        run("fp16 max %d" % n, np.full((n, K), 65504.0, np.float16), ci, np.full((q, K), 3.0, np.float32), qi, k)
        run("subnormal %d" % n, np.full((n, K), 6e-8, np.float16), ci, qvr, qi, k)
        run("all negative %d" % n, -cvr, ci, qvr, qi, k)
        run("negative queries %d" % n, cvr, ci, -qvr, qi, k)
        one = cvr.copy()
        one[:, 1:] = 0
        run("one column %d" % n, one, ci, qvr, qi, k)
    print("all degenerate cases ok")


if __name__ == "__main__":
    main()
