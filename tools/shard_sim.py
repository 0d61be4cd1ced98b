#!/usr/bin/env python
"""Estimate the N-GPU step time on ONE GPU: build the N row shards of the bench corpus one after the
other on the same device, run the staged search exactly like dist.sharded_search would (collectives
replaced by in-process tensor ops) and report the slowest shard's begin / finish time."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--workload", default="hybrid")
    ap.add_argument("--n-docs", type=int, default=8_841_823)
    ap.add_argument("--n-queries", type=int, default=6980)
    ap.add_argument("--first-rows", type=int, default=0)
    ap.add_argument("--max-growth", type=int, default=0)
    ap.add_argument("--main-chunks", type=int, default=0)
    ap.add_argument("--only", type=int, default=0, help="build only the first N shards of the partition (lets the default list capacity fit one GPU)")
    ap.add_argument("--cand-cap", type=int, default=65536)
    ap.add_argument("--sample-period", type=int, default=-1)
    ap.add_argument("--overlap-aux", type=int, default=-1)
    ap.add_argument("--aux-cus", type=int, default=-1)
    ap.add_argument("--mid", type=int, default=0, help="1: second threshold agreement after the first slice of the main pass (dhr_search_mid)")
    ap.add_argument("--pre", type=int, default=1, help="1 (default): the first agreement in two rounds (dhr_search_pre / dhr_search_begin_rest) where the shards offer it; 0: dhr_search_begin")
    a = ap.parse_args()
    import torch
    import bench
    from dhr_amd import dist as D, synth, _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    d_dlr, d_cls = (768, 768) if a.workload == "hybrid" else (0, 768)
    k = 1000
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, a.n_queries, d_dlr, d_cls, 4, 12, False)
    # per-shard: begin; keep sample; destroy? the finish needs the handle -> keep all shards resident (fits: ~13 GB each)
    shards = []
    for r in range(a.only or a.shards):
        lo, hi = D.shard_bounds(a.n_docs, a.shards, r)
        cv, ci = bench.gen_rows(torch, synth, dev, 1237, lo, hi, d_dlr, d_cls, 30, 90, False)
        ix = GipIndex(cv, ci, row_offset=lo)
        ix.set_param(_lib.PARAM_PROFILE, 1)
        if a.cand_cap:
            ix.set_param(_lib.PARAM_CAND_CAP, a.cand_cap)
        if a.main_chunks:
            ix.set_param(_lib.PARAM_MAIN_CHUNKS, a.main_chunks)
        if a.max_growth:
            ix.set_param(_lib.PARAM_MAX_GROWTH, a.max_growth)
        if a.first_rows:
            ix.set_param(_lib.PARAM_FIRST_ROWS, a.first_rows)
        if a.sample_period >= 0:
            ix.set_param(_lib.PARAM_SAMPLE_PERIOD, a.sample_period)
        if a.overlap_aux >= 0:
            ix.set_param(_lib.PARAM_OVERLAP_AUX, a.overlap_aux)
        if a.aux_cus >= 0:
            ix.set_param(_lib.PARAM_AUX_CUS, a.aux_cus)
        shards.append(ix)
        del cv, ci
        torch.cuda.empty_cache()
    for ix in shards:
        ix.set_param(_lib.PARAM_SAMPLE_SHARE, a.shards)
    rnk = shards[0].union_rank(k)
    results = {}
    for mid in ([0, 1] if a.mid else [0]):
        for it in range(2):
            tb, tf, samples, outs = [], [], [], []
            rl_pre, ru_pre = max(s.pre_ranks(k)[0] for s in shards), max(s.pre_ranks(k)[1] for s in shards)
            if a.pre and rl_pre > 0:
                firsts, tb2 = [], []
                for ix in shards:
                    torch.cuda.synchronize(); t = time.perf_counter()
                    firsts.append(ix.search_pre(qv, qi, k, rl_pre)); torch.cuda.synchronize(); tb.append(time.perf_counter() - t)
                t = time.perf_counter(); tau0 = D.common_threshold(torch.stack(firsts), min(ru_pre, len(shards) * rl_pre)); torch.cuda.synchronize(); t_pre = time.perf_counter() - t
                for ix in shards:
                    torch.cuda.synchronize(); t = time.perf_counter()
                    samples.append(ix.search_begin_rest(tau0)); torch.cuda.synchronize(); tb2.append(time.perf_counter() - t)
                if it == 1:
                    print("  begin in two rounds: first part max %.2f ms + threshold %.2f ms + rest max %.2f ms (ranks local %d / union %d)" % (max(tb) * 1e3, t_pre * 1e3, max(tb2) * 1e3, rl_pre, ru_pre))
                tb = [x + y + t_pre for x, y in zip(tb, tb2)]
            else:
                for ix in shards:
                    torch.cuda.synchronize(); t = time.perf_counter()
                    samples.append(ix.search_begin(qv, qi, k)); torch.cuda.synchronize(); tb.append(time.perf_counter() - t)
            st_begin = shards[0].stats()
            t = time.perf_counter(); tau = D.common_threshold(torch.stack(samples), rnk); torch.cuda.synchronize(); tt = time.perf_counter() - t
            tmid, tt2 = [0.0], 0.0
            if mid:
                rl, ru = shards[0].mid_ranks(k)
                mids, tmid = [], []
                for ix in shards:
                    torch.cuda.synchronize(); t = time.perf_counter()
                    mids.append(ix.search_mid(tau)); torch.cuda.synchronize(); tmid.append(time.perf_counter() - t)
                t = time.perf_counter(); tau = torch.maximum(tau, D.common_threshold(torch.stack(mids), ru)); torch.cuda.synchronize(); tt2 = time.perf_counter() - t
            for ix in shards:
                torch.cuda.synchronize(); t = time.perf_counter()
                outs.append(ix.search_finish(tau)); torch.cuda.synchronize(); tf.append(time.perf_counter() - t)
            t = time.perf_counter()
            cnts = torch.stack([o[2] for o in outs]); kk = min(k, (int(cnts.max()) + 63) // 64 * 64)
            gs = torch.stack([o[0][:, :kk] for o in outs]); gr = torch.stack([o[1][:, :kk] for o in outs])     # what the all-gather leaves
            torch.cuda.synchronize(); t = time.perf_counter()
            ms, mr = D.merge_sorted_lists(gs, gr, k); torch.cuda.synchronize(); tm = time.perf_counter() - t
            tot = torch.stack([o[2] for o in outs]).clamp(min=0).sum(0)
            call = torch.stack([o[2] for o in outs])
            fmask = (tot < k) | (call < 0).any(0)
            failed = int(fmask.sum())
            if it == 1 and failed:
                for q in torch.nonzero(fmask).flatten().tolist()[:8]:
                    print("  failed query %d: counts per shard %s  sum %d  tau %.5f  own 26th-best per shard %s" %
                          (q, call[:, q].tolist(), int(tot[q]), float(tau[q]), [round(float(sm[q, -1]), 5) for sm in samples]))
        st = shards[0].stats()
        print("shards %d rank r=%d : begin max %.2f ms  mid max %.2f ms + tau2 %.2f ms  finish max %.2f ms  tau %.2f ms  merge %.2f ms  -> est. step %.2f ms (+ all-gather)  failed queries %d"
              % (a.shards, rnk, max(tb) * 1e3, max(tmid) * 1e3, tt2 * 1e3, max(tf) * 1e3, tt * 1e3, tm * 1e3, (max(tb) + max(tmid) + tt2 + max(tf) + tt + tm) * 1e3, failed))
        if mid:
            print("mid ranks (local, union):", shards[0].mid_ranks(k))
        results[mid] = (ms, mr, fmask)
    if a.mid:
        ok = ~(results[0][2] | results[1][2])
        print("mid == plain on the %d queries neither protocol flagged: rows %s scores %s" % (int(ok.sum()), bool((results[0][1][ok] == results[1][1][ok]).all()),
              bool((results[0][0][ok] == results[1][0][ok]).all())))
    print("shard0 stats after begin:", {k_: (round(v, 2) if isinstance(v, float) else v) for k_, v in st_begin.items()})
    print("shard0 stats:", {k_: (round(v, 2) if isinstance(v, float) else v) for k_, v in st.items()})


if __name__ == "__main__":
    main()
