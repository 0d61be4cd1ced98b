#!/usr/bin/env python
"""Where do the exact rescorings of a config-3 step come from?  Separates "the threshold arrives late" (controller) from "the bound is
loose" (operand image): the same batch is searched (a) normally, (b) staged with the sampled run's own thresholds, (c) staged with every
query's TRUE k-th best score as the threshold of the whole main pass -- the floor of what any controller could reach with this bound.
Prints bound / exact candidates per query and the phase times of each."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8841823)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--cls", type=int, default=768)
    args = ap.parse_args()
    import torch
    import bench
    from dhr_amd import synth, _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    n, nq, k = args.rows, args.queries, args.k
    cv, ci = bench.gen_rows(torch, synth, dev, 1237, 0, n, 768, args.cls, 30, 90, False)
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, nq, 768, args.cls, 4, 12, False)
    ix = GipIndex(cv, ci, device=0)
    del cv, ci
    torch.cuda.empty_cache()
    ix.set_param(_lib.PARAM_PROFILE, 1)

    def brief(st=None):
        st = st or ix.stats()
        return ("phases %2d gemm %6.1f refine %5.1f rescore %5.1f select %4.1f total %6.1f ms | bound %7.0f exact %6.0f per query, redone %d"
                % (st["phases"], st["gemm_ms"], st["refine_ms"], st["rescore_ms"], st["select_ms"], st["total_ms"], st["candidates_bound"] / nq,
                   st["candidates_exact"] / nq, st["sample_fallback_queries"]))

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, out

    for ov in (0, 1, 0, 1):
        ix.set_param(_lib.PARAM_OVERLAP_AUX, ov)
        ms, (s0, r0) = timed(lambda: ix.search(qv, qi, k, out_device=True))
        print("search overlap_aux=%d: %.1f ms wall | %s" % (ov, ms, brief()), flush=True)
    ix.set_param(_lib.PARAM_OVERLAP_AUX, -1)
    tau_true = s0[:, k - 1].contiguous()

    def staged(tau_fn, label, overlap):
        ix.set_param(_lib.PARAM_OVERLAP_AUX, overlap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        smp = ix.search_begin(qv, qi, k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st_b = ix.stats()
        tau = tau_fn(smp)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        s, r, c = ix.search_finish(tau)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        st_f = ix.stats()
        same = bool(torch.equal(r, r0)) and bool(torch.equal(s, s0))
        print("%s (overlap %d): begin %.1f ms finish %.1f ms, identical to dhr_search: %s" % (label, overlap, (t1 - t0) * 1e3, (t3 - t2) * 1e3, same))
        print("    after begin : %s" % brief(st_b))
        print("    after finish: %s" % brief(st_f), flush=True)
    rr = ix.sample_rank(k)
    for ov in (0, 1):
        staged(lambda smp: smp[:, rr - 1].contiguous(), "staged, sampled thresholds (rank %d of the sample), no extrapolation" % rr, ov)
        staged(lambda smp: tau_true, "staged, TRUE k-th best as the threshold (floor)", ov)
    ix.close()


if __name__ == "__main__":
    main()
