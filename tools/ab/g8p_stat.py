#!/usr/bin/env python
"""Hit-stack statistics of the persistent bound GEMM (-DG8_TRACE=1 build): over one full search of the bench workload, how many
(wave, tile) scans hold a lane with more than D hits (= would overflow a private stack of depth D).  Timing tool, not a test."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    import torch
    import bench
    from dhr_amd import _lib, synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    cv, ci = bench.gen_shard(torch, synth, dev, 4242, rows, 768, 768, 30, 90, False)
    qv, qi = bench.gen_shard(torch, synth, dev, 777, 6980, 768, 768, 4, 12, False)
    ix = GipIndex(cv, ci)
    del cv
    st = getattr(ix._lib, "dhr_debug_g8p_stat")
    st.argtypes = [C.c_void_p]
    buf = np.zeros(16, dtype=np.uint64)
    ix.search(qv, qi, 1000, out_device=True)
    torch.cuda.synchronize()
    assert st(buf.ctypes.data) == 0             # warm-up search dropped
    ix.search(qv, qi, 1000, out_device=True)
    torch.cuda.synchronize()
    assert st(buf.ctypes.data) == 0
    n = float(buf[0])
    print("wave-tiles scanned %d, hits %d (%.2f per wave-tile, %.4f %% of the accumulators), lanes with a hit per wave-tile %.2f" %
          (buf[0], buf[10], buf[10] / n, 100.0 * buf[10] / (n * 8192), buf[11] / n))
    for i, d in enumerate((3, 5, 6, 8, 10, 13, 16, 32)):
        print("  some lane holds more than %2d hits: %.4f of the wave-tiles" % (d, buf[1 + i] / n))
    print(ix.stats())
    ix.close()


if __name__ == "__main__":
    main()
