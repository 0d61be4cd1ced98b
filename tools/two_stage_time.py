#!/usr/bin/env python
"""Times the two-stage modes of the reference's docs (--theta 0.3 --rerank, --IP --rerank; gip_retrieval.py:128-156) on the config-3
corpus: dhr_search_rerank with agip_topk = 10 000 candidates, top-1000 after the exact rerank, against the brute-force search."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""          # "theta0.3" | "theta0.1" | "ip": that mode alone (kernel traces)
    import torch
    import bench
    from dhr_amd import synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    n, nq, k, k1 = 8841823, 6980, 1000, 10000
    cv, ci = bench.gen_rows(torch, synth, dev, 1237, 0, n, 768, 768, 30, 90, False)
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, nq, 768, 768, 4, 12, False)
    ix = GipIndex(cv, ci, device=0)
    del cv, ci
    torch.cuda.empty_cache()
    q = qv.float()

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out
    from dhr_amd import _lib
    ix.set_param(_lib.PARAM_PROFILE, 1)
    if os.environ.get("TS_GROWTH"):
        ix.set_param(_lib.PARAM_MAX_GROWTH, int(os.environ["TS_GROWTH"]))

    def brief():
        st = ix.stats()
        return ("phases %d  gemm %.1f refine %.1f rescore %.1f select %.1f total %.1f ms  bound %.0f exact %.0f per query  redone %d  overflow retries %d"
                % (st["phases"], st["gemm_ms"], st["refine_ms"], st["rescore_ms"], st["select_ms"], st["total_ms"], st["candidates_bound"] / nq,
                   st["candidates_exact"] / nq, st["sample_fallback_queries"], st["overflow_retries"]))
    ms_b, (sb, rb) = timed(lambda: ix.search(q, qi, k, out_device=True), reps=1 if only else 3)
    print("brute force (theta 0)          : %.1f ms per 6 980 queries" % ms_b)
    print("   ", brief())
    for theta in (0.3, 0.1):
        if only and only != "theta%.1f" % theta:
            continue
        q1 = torch.where(q > theta, q, torch.zeros_like(q))
        ms, (s2, r2) = timed(lambda: ix.search_rerank(q1, qi, q, qi, k1, k))
        rec = np.mean([len(set(r2[i].tolist()) & set(rb[i].cpu().tolist())) / k for i in range(0, nq, 349)])
        print("theta %.1f --rerank (agip 10000)  : %.1f ms (batches resident on the device, result lists copied to the host), overlap with brute force top-1000: %.3f" % (theta, ms, rec))
        print("   ", brief())
        nz = (q1 > 0).sum(1).float()
        print("    non-zero stage-1 query columns: mean %.1f max %d; gated %.1f dense %.1f" % (nz.mean(), int(nz.max()), (q1[:, :768] > 0).sum(1).float().mean(), (q1[:, 768:] > 0).sum(1).float().mean()))
    if only and only != "ip":
        ix.close()
        return
    ms, (s2, r2) = timed(lambda: ix.search_rerank(q, None, q, qi, k1, k))
    rec = np.mean([len(set(r2[i].tolist()) & set(rb[i].cpu().tolist())) / k for i in range(0, nq, 349)])
    print("--IP --rerank (agip 10000)      : %.1f ms, overlap %.3f" % (ms, rec))
    print("   ", brief())
    ix.close()


if __name__ == "__main__":
    main()
