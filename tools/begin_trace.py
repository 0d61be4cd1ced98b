#!/usr/bin/env python
"""One 1/8 shard of the bench corpus: dhr_search_pre + dhr_search_begin_rest (the two-round begin of the sharded search) a few times, for a
kernel timeline (run under rocprofv3 --kernel-trace, then tools/timeline.py on the database).  The thresholds of the rest come from the shard's own
first part (one shard cannot form the union; the timeline of the kernels is what is wanted)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import bench
    from dhr_amd import dist as D, synth, _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    k, ns = 1000, 8
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, 6980, 768, 768, 4, 12, False)
    lo, hi = D.shard_bounds(8_841_823, ns, 0)
    cv, ci = bench.gen_rows(torch, synth, dev, 1237, lo, hi, 768, 768, 30, 90, False)
    ix = GipIndex(cv, ci, row_offset=lo)
    del cv, ci
    ix.set_param(_lib.PARAM_SAMPLE_SHARE, ns)
    rl, ru = ix.pre_ranks(k)
    for _ in range(3):
        first = ix.search_pre(qv, qi, k, rl)
        tau0 = first[:, min(rl, 4) - 1].contiguous()          # a stand-in for the union's threshold: this shard's 4th best so far
        torch.cuda.synchronize()
        sample = ix.search_begin_rest(tau0)
        torch.cuda.synchronize()
        tau = sample[:, 7].contiguous()
        ix.search_finish(tau)
        torch.cuda.synchronize()
    ix.close()


if __name__ == "__main__":
    main()
