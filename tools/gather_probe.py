#!/usr/bin/env python
"""Does the row-gather rate of the rescoring kernel depend on how far apart the rows lie (TLB reach / page locality)?  dhr_score_rows of
Q x m random rows drawn from windows of decreasing size of one resident index; GB/s of row bytes per window.
usage: python tools/gather_probe.py [dense|hybrid]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
    import torch
    import bench
    from dhr_amd import synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    n, nq, m = 8_841_823, 6980, 1536
    d_dlr = 0 if kind == "dense" else 768
    cv, ci = bench.gen_rows(torch, synth, dev, 1237, 0, n, d_dlr, 768, 30, 90, False)
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, nq, d_dlr, 768, 4, 12, False)
    ix = GipIndex(cv, ci, device=0)
    del cv, ci
    torch.cuda.empty_cache()
    row_bytes = (d_dlr + 768) * 2 + d_dlr
    g = torch.Generator(device=dev).manual_seed(3)
    for win in (n, 4_000_000, 2_000_000, 700_000, 200_000, 50_000):
        lo = (n - win) // 2
        rows = torch.randint(lo, lo + win, (nq, m), generator=g, device=dev, dtype=torch.int64)
        ix.score_rows_device(qv, qi, rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            ix.score_rows_device(qv, qi, rows)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("%s: %d x %d random rows from a window of %9d rows (%7.1f MB): %.2f ms = %.2f TB/s of row bytes"
              % (kind, nq, m, win, win * row_bytes / 1e6, dt * 1e3, nq * m * row_bytes / dt / 1e12), flush=True)
    # the same rows, each query's list sorted by row
    rows = torch.randint(0, n, (nq, m), generator=g, device=dev, dtype=torch.int64)
    srt = torch.sort(rows, dim=1).values
    for name, r in (("unsorted", rows), ("sorted per query", srt)):
        ix.score_rows_device(qv, qi, r)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            ix.score_rows_device(qv, qi, r)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print("%s: whole corpus, %s: %.2f ms = %.2f TB/s" % (kind, name, dt * 1e3, nq * m * row_bytes / dt / 1e12), flush=True)
    ix.close()


if __name__ == "__main__":
    main()
