#!/usr/bin/env python
"""Controller A/B at config 3 (or --rows / --cls): one index, several parameter settings, wall time + phase times + candidate counts
of each, all results compared bit for bit with the first.  A/B builds of the library (tools/ab_build.sh, DHR_HIP_LIB) need a
process of their own: run this script once per setting."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=8841823)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--k", type=int, default=1000)
    ap.add_argument("--dlr", type=int, default=768)
    ap.add_argument("--cls", type=int, default=768)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--sets", default="", help="semicolon-separated settings, each 'name=value,name=value' of overlap,period,chunks,growth,aux_cus,prog")
    ap.add_argument("--staged", action="store_true", help="also print the begin / finish split (dhr_search_begin / finish with the sample's own threshold)")
    args = ap.parse_args()
    import torch
    import bench
    from dhr_amd import synth, _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    n, nq, k = args.rows, args.queries, args.k
    cv, ci = bench.gen_rows(torch, synth, dev, 1237, 0, n, args.dlr, args.cls, 30, 90, False)
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, nq, args.dlr, args.cls, 4, 12, False)
    ix = GipIndex(cv, ci, device=0)
    del cv, ci
    torch.cuda.empty_cache()
    ix.set_param(_lib.PARAM_PROFILE, 1)
    names = {"overlap": _lib.PARAM_OVERLAP_AUX, "period": _lib.PARAM_SAMPLE_PERIOD, "chunks": _lib.PARAM_MAIN_CHUNKS, "growth": _lib.PARAM_MAX_GROWTH,
             "aux_cus": _lib.PARAM_AUX_CUS, "prog": _lib.PARAM_PROGRESSIVE_THR}
    defaults = {"overlap": -1, "period": 32, "chunks": 2, "growth": 32, "aux_cus": -1, "prog": 2}

    def brief(st):
        return ("phases %2d gemm %6.1f refine %5.1f rescore %5.1f select %4.1f total %6.1f ms | bound %7.0f exact %6.0f per query, redone %d"
                % (st["phases"], st["gemm_ms"], st["refine_ms"], st["rescore_ms"], st["select_ms"], st["total_ms"], st["candidates_bound"] / nq,
                   st["candidates_exact"] / nq, st["sample_fallback_queries"]))
    ref = None
    print("env: DHR_HIP_LIB=%s" % os.environ.get("DHR_HIP_LIB"), flush=True)
    for setting in ([""] + [x for x in args.sets.split(";") if x]):
        cur = dict(defaults)
        for kv in [x for x in setting.split(",") if x]:
            a, b = kv.split("=")
            cur[a] = int(b)
        for a, v in cur.items():
            if a == "aux_cus" and v < 0:
                continue
            ix.set_param(names[a], v)
        ix.search(qv, qi, k, out_device=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            s, r = ix.search(qv, qi, k, out_device=True)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.reps * 1e3
        if ref is None:
            ref = (s.clone(), r.clone())
        same = bool(torch.equal(s, ref[0])) and bool(torch.equal(r, ref[1]))
        print("[%-28s] %.2f ms | %s | identical %s" % (setting or "defaults", ms, brief(ix.stats()), same), flush=True)
    if args.staged:
        for a, v in defaults.items():
            if not (a == "aux_cus" and v < 0):
                ix.set_param(names[a], v)
        rr = ix.sample_rank(k)
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            smp = ix.search_begin(qv, qi, k)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            st_b = ix.stats()
            tau = smp[:, rr - 1].contiguous()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            s, r, c = ix.search_finish(tau)
            torch.cuda.synchronize(); t3 = time.perf_counter()
        print("staged: begin %.2f ms finish %.2f ms identical %s" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, bool(torch.equal(r, ref[1]))))
        print("    after begin : %s" % brief(st_b))
        print("    after finish: %s" % brief(ix.stats()), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
