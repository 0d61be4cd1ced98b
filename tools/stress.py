#!/usr/bin/env python
"""Randomised differential test: many small random configurations (shapes, dtypes, value signs, fp32 / fp16 queries, bucket
counts, k, gated / ungated, index parameters) through the C ABI against the oracle's float64 scores.  Prints the failing
configuration and exits non-zero on the first mismatch.  usage: python tools/stress.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from dhr_amd import _lib
    from dhr_amd.retrieval import gip_retrieval as G
    from oracle import gip_oracle as O
    rng = np.random.default_rng(seed)
    t0 = time.time()
    for case in range(n_cases):
        n = int(rng.choice([37, 256, 300, 1000, 2500, 5000, 20000]))
        q = int(rng.integers(1, 20))
        d_dlr = int(rng.choice([0, 8, 32, 64, 96, 128, 256, 768]))
        d_cls = int(rng.choice([0, 8, 24, 64, 128, 768]))
        if d_dlr + d_cls == 0:
            d_cls = 64
        k = int(min(n, rng.choice([1, 5, 10, 100, 1000])))
        idx_dtype = rng.choice([np.uint8, np.int8, np.int16])
        n_idx = int(rng.choice([2, 5, 39, 120]))
        neg = bool(rng.random() < 0.3)
        q32 = bool(rng.random() < 0.4)
        sparse_vals = bool(rng.random() < 0.5)
        nb = int(rng.choice([0, 0, 0, 1, 2]))
        ungated = bool(d_dlr and rng.random() < 0.15)
        g8 = int(rng.integers(0, 2))                   # image of the gated half: fp16 2:4 (the library's choice at these sizes) or int8 2:4
        os.environ["DHR_GATED_I8"] = str(g8)
        cfg = dict(case=case, n=n, q=q, d_dlr=d_dlr, d_cls=d_cls, k=k, idx=np.dtype(idx_dtype).name, n_idx=n_idx, neg=neg, q32=q32,
                   sparse=sparse_vals, nb=nb, ungated=ungated, gated_i8=g8)
        K = d_dlr + d_cls
        def vals(m):
            v = np.abs(rng.standard_normal((m, K))) * 0.5
            if sparse_vals and d_dlr:
                v[:, :d_dlr] *= (rng.random((m, d_dlr)) < 0.1)
                v[:, :d_dlr] += rng.random((m, d_dlr)) * 0.02
            if neg and d_dlr:
                v[:, :d_dlr] *= np.where(rng.random((m, d_dlr)) < 0.2, -1.0, 1.0)
            v[:, d_dlr:] = rng.standard_normal((m, d_cls)) * 0.1
            return v
        cv = vals(n).astype(np.float16)
        qv = vals(q)
        qv = (qv * 0.37).astype(np.float32) if q32 else qv.astype(np.float16).astype(np.float32)
        if d_dlr:
            lo = -n_idx // 2 if np.dtype(idx_dtype).kind == "i" else 0
            ci = rng.integers(lo, lo + n_idx, (n, d_dlr)).astype(idx_dtype)
            # the query index may have another dtype and a wider range than the corpus index (torch promotes the compare)
            q_dtype = rng.choice([np.uint8, np.int8, np.int16])
            qlo = -n_idx // 2 if np.dtype(q_dtype).kind == "i" else 0
            qi = rng.integers(qlo, qlo + n_idx + (40 if np.dtype(q_dtype).itemsize == 2 else 0), (q, d_dlr)).astype(q_dtype)
            cfg["q_idx"] = np.dtype(q_dtype).name
        else:
            ci = qi = None
        try:
            ix = G.GipIndex(cv, ci, idx_buckets=nb if d_dlr else 0)
            if rng.random() < 0.3:
                ix.set_param(_lib.PARAM_CAND_CAP, int(rng.choice([1024, 4096])))
            qidx = None if (ungated or ci is None) else qi
            s, r = ix.search(qv, qidx, k)
            ix.close()
            c32 = cv.astype(np.float32)
            for i in range(q):
                ex = O.gip_scores_f64(qv[i], None if qidx is None else qidx[i], c32, None if qidx is None else ci)
                O.check_topk(r[i], s[i], ex, k)
                np.testing.assert_allclose(s[i], ex[r[i]].astype(np.float32), rtol=0, atol=1e-6 * max(1.0, np.abs(ex).max()))
        except Exception as e:  # noqa: BLE001
            print("FAILED", cfg, "->", repr(e)[:500])
            sys.exit(1)
        if case % 25 == 0:
            print("case %d ok (%.0f s) %s" % (case, time.time() - t0, cfg), flush=True)
    print("all %d cases ok in %.0f s" % (n_cases, time.time() - t0))


if __name__ == "__main__":
    main()
