#!/usr/bin/env python
"""The unsharded search of the bench corpus (config 3, or --dense: config 2) three times, for a kernel timeline of ONE step: run under
rocprofv3 --kernel-trace, then tools/timeline.py <db> <kernels per step>.  usage: python tools/step_trace.py [--dense]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    import torch
    import bench
    from dhr_amd import synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dense = "--dense" in sys.argv
    dev = torch.device("cuda", 0)
    d_dlr = 0 if dense else 768
    qv, qi = bench.gen_rows(torch, synth, dev, 1237 + 999_983, 0, 6980, d_dlr, 768, 4, 12, False)
    cv, ci = bench.gen_rows(torch, synth, dev, 1237, 0, 8_841_823, d_dlr, 768, 30, 90, False)
    ix = GipIndex(cv, ci)
    del cv, ci
    for _ in range(3):
        ix.search(qv, qi, 1000, out_device=True)
        torch.cuda.synchronize()
    print(ix.stats())
    ix.close()


if __name__ == "__main__":
    main()
