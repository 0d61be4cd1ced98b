#!/usr/bin/env python
"""Experiment: what would ordering the queries of a batch by their candidate rate be worth?  
The rate is estimated from the bound scores of a 16 k-row sample against the final k-th scores; the same search is then timed
with the queries in their own order, in ascending rate order and in a random order."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    import torch
    import bench
    from dhr_amd import _lib, synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    n, nq, k = 8841823, 6980, 1000
    cv, ci = bench.gen_rows(torch, synth, dev, 1237, 0, n, 768, 768, 30, 90, False)
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, nq, 768, 768, 4, 12, False)
    ix = GipIndex(cv, ci, device=0)
    del cv, ci
    torch.cuda.empty_cache()
    ix.set_param(_lib.PARAM_PROFILE, 1)
    s, r = ix.search(qv, qi, k, out_device=True)
    kth = s[:, k - 1].clone()
    qb, keep = _lib.make_query_batch(qv, qi)
    m = 16384
    out = torch.zeros((nq, m), dtype=torch.float32, device=dev)
    _lib.check(ix._lib.dhr_debug_bound_scores(ix._h, C.byref(qb), 1_000_000, 1_000_000 + m, out.data_ptr(), 0), "debug_bound")
    rate = (out >= (kth[:, None] - 0.2)).sum(1).float() / m
    print("candidate rate per row: mean %.5f median %.5f max %.4f" % (rate.mean().item(), rate.median().item(), rate.max().item()))
    orders = {"own": torch.arange(nq, device=dev), "ascending rate": torch.argsort(rate), "random": torch.randperm(nq, device=dev)}

    def timed(perm, reps=6):
        q2, i2 = qv[perm].contiguous(), qi[perm].contiguous()
        ix.search(q2, i2, k, out_device=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = 0.0
        for _ in range(reps):
            s2, r2 = ix.search(q2, i2, k, out_device=True)
            g += ix.stats()["gemm_ms"]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps * 1e3
        assert torch.equal(r2, r[perm])
        return dt, g / reps
    for rnd in range(2):
        for name, perm in orders.items():
            dt, g = timed(perm)
            print("%-15s %.2f ms per step, GEMM %.2f ms" % (name, dt, g))
    ix.close()


if __name__ == "__main__":
    main()
