#!/usr/bin/env python
"""The two forms of the fp16 2:4 bound GEMM (variant 4: 4 waves of 128 x 128, variant 5: 8 waves of 128 x 64) against each other: bound scores
bit-identical on a ragged shard (gated and ungated batches), stand-alone timing of both on the bench's synthetic data (closed filter), identical
search results.  DHR_GATED_I8=0 keeps the fp16 image (the library picks the integer image, which has one kernel, for indexes of this size)."""
import ctypes as C
import os
import sys

os.environ.setdefault("DHR_GATED_I8", "0")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
from dhr_amd import _lib, synth
from dhr_amd.retrieval.gip_retrieval import GipIndex


def bound(ix, qv, qi, n, variant):
    ix.set_param(_lib.PARAM_GEMM_VARIANT, variant)
    qb, keep = _lib.make_query_batch(qv, qi)
    out = torch.zeros((qv.shape[0], n), dtype=torch.float32, device="cuda")
    _lib.check(ix._lib.dhr_debug_bound_scores(ix._h, C.byref(qb), 0, n, out.data_ptr(), 0), "debug_bound")
    return out.cpu().numpy()


def main():
    for (n, q, d_cls) in ((1000, 40, 128), (5000, 300, 768)):
        cv, ci, qv, qi = synth.make_pair(3, n, q, 768, d_cls)
        ix = GipIndex(cv, ci, idx_buckets=2)
        q32 = qv.astype(np.float32)
        u3 = bound(ix, q32, qi, n, 5)
        u3i = bound(ix, q32, None, n, 5)
        for v in (4,):
            u4 = bound(ix, q32, qi, n, v)
            u4i = bound(ix, q32, None, n, v)          # ungated batch: 2 * ts sparse stages on the query side
            print("n %d q %d d_cls %d: variant %d gated max|v-v5| %.3g  ungated %.3g  (|u| max %.3f)" % (n, q, d_cls, v, np.abs(u4 - u3).max(), np.abs(u4i - u3i).max(), np.abs(u3).max()))
            assert np.array_equal(u4, u3) and np.array_equal(u4i, u3i)
        ix.close()
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    import bench
    dev = torch.device("cuda", 0)
    cv, ci = bench.gen_shard(torch, synth, dev, 4242, rows, 768, 768, 30, 90, False)
    qv, qi = bench.gen_shard(torch, synth, dev, 777, 6980, 768, 768, 4, 12, False)
    ix = GipIndex(cv, ci, idx_buckets=2)
    del cv
    qb, keep = _lib.make_query_batch(qv, qi)
    for rep in range(2):
        for variant in (4, 5):
            ix.set_param(_lib.PARAM_GEMM_VARIANT, variant)
            ms, fl = C.c_double(), C.c_double()
            _lib.check(ix._lib.dhr_debug_gemm_time(ix._h, C.byref(qb), 8, C.byref(ms), C.byref(fl), None), "gemm_time")
            alg = 2.0 * rows * 6980 * 1536
            print("variant %d: %.3f ms per %d rows, algorithmic %.1f TFLOP/s (frac %.3f)" % (variant, ms.value, rows, alg / ms.value / 1e9, alg / ms.value / 1e9 / 2500))
    # a search with each variant: identical results
    res = []
    for variant in (4, 5):
        ix.set_param(_lib.PARAM_GEMM_VARIANT, variant)
        s, r = ix.search(qv, qi, 1000, out_device=True)
        res.append((s.cpu(), r.cpu()))
    for o in res[1:]:
        assert torch.equal(res[0][0], o[0]) and torch.equal(res[0][1], o[1])
    print("search results identical")
    ix.close()


if __name__ == "__main__":
    main()
