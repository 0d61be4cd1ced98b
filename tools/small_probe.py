#!/usr/bin/env python
"""One small corpus (BEIR sizes), a few searches, wall + event time of each: finds host-side stalls of the small-corpus controller.
usage: python tools/small_probe.py ROWS QUERIES [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    n, nq = int(sys.argv[1]), int(sys.argv[2])
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    import torch
    import bench
    from dhr_amd import synth, _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    cv, ci = bench.gen_rows(torch, synth, dev, 1242, 0, n, 768, 128, 30, 90, False)
    qv, qi = bench.gen_rows(torch, synth, dev, 1242 + 999_983, 0, nq, 768, 128, 4, 12, False)
    ix = GipIndex(cv, ci, device=0)
    ix.set_param(_lib.PARAM_PROFILE, int(os.environ.get("PROFILE", "1")))
    k = min(1000, n)
    for i in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, r = ix.search(qv, qi, k, out_device=True)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = ix.stats()
        print("rows %d queries %d: wall %.3f ms | total %.3f gemm %.3f refine %.3f rescore %.3f select %.3f prep %.3f | phases %d redone %d retries %d"
              % (n, nq, (t1 - t0) * 1e3, st["total_ms"], st["gemm_ms"], st["refine_ms"], st["rescore_ms"], st["select_ms"], st["prep_ms"], st["phases"],
                 st["sample_fallback_queries"], st["overflow_retries"]), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
