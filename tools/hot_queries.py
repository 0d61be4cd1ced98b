#!/usr/bin/env python
"""How skewed are the bound-candidate counts over the queries?  Bound scores U of a 100k-row slice against all bench
queries, counted against each query's final k-th best exact score (the filter threshold up to the margin)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    import torch, bench
    from dhr_amd import _lib, synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    n, nq, k, m = 2_000_000, 6980, 1000, 100_000
    cv, ci = bench.gen_shard(torch, synth, dev, 1237, n, 768, 768, 30, 90, False)
    qv, qi = bench.gen_shard(torch, synth, dev, 1237 + 999_983, nq, 768, 768, 4, 12, False)
    ix = GipIndex(cv, ci)
    s, r = ix.search(qv, qi, k, out_device=True)
    kth = s[:, k - 1]
    qb, keep = _lib.make_query_batch(qv, qi)
    U = torch.empty((nq, m), dtype=torch.float32, device=dev)
    _lib.check(ix._lib.dhr_debug_bound_scores(ix._h, C.byref(qb), 0, m, U.data_ptr(), None), "bound_scores")
    cnt = (U >= kth[:, None]).sum(1).float() * (n / m)          # extrapolated to the 2M-row corpus
    c = cnt.cpu().numpy()
    order = np.sort(c)[::-1]
    tot = c.sum()
    print("bound candidates per query (2M rows, k-th-score threshold): mean %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f" %
          (c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max()))
    for frac in (0.001, 0.01, 0.05, 0.1, 0.25):
        top = int(nq * frac)
        print("  hottest %5.1f %% of the queries hold %5.1f %% of all bound candidates" % (frac * 100, 100 * order[:top].sum() / tot))
    # what distinguishes the hot ones
    qn = qv[:, :768].float()
    l1 = qn.sum(1).cpu().numpy(); heavy = (qn > 0.05).sum(1).cpu().numpy()
    hot = np.argsort(-c)[:70]
    print("  hot queries: gated L1 mass %.2f (all %.2f), heavy slices %.1f (all %.1f), k-th score %.3f (all %.3f)" %
          (l1[hot].mean(), l1.mean(), heavy[hot].mean(), heavy.mean(), kth.cpu().numpy()[hot].mean(), float(kth.mean())))
    mg = np.zeros(nq, np.float32)
    _lib.check(ix._lib.dhr_debug_query_margins(ix._h, C.byref(qb), mg.ctypes.data, None), "margins")
    cnt_m = ((U >= (kth - torch.from_numpy(mg).to(dev))[:, None]).sum(1).float() * (n / m)).cpu().numpy()
    print("  with the filter margin: mean %.0f  p99 %.0f  max %.0f;  margins: mean %.3f  max %.3f" % (cnt_m.mean(), np.percentile(cnt_m, 99), cnt_m.max(), mg.mean(), mg.max()))
    qg = qv[:, :768].float().cpu().numpy(); qd = qv[:, 768:].float().cpu().numpy()
    kk = kth.cpu().numpy()
    for q in np.argsort(-cnt_m)[:8]:
        print("    query %4d: %8.0f cand. (%.0f without margin)  margin %.3f  k-th %.3f  gated max %.3f  terms>0.05 %d  gated L1 %.2f  ungated norm %.3f max %.3f" %
              (q, cnt_m[q], c[q], mg[q], kk[q], qg[q].max(), int((qg[q] > 0.05).sum()), qg[q].sum(), np.linalg.norm(qd[q]), np.abs(qd[q]).max()))
    ix.close()


if __name__ == "__main__":
    main()
