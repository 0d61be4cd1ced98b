#!/usr/bin/env python
"""Stand-alone timing of the bound GEMM variants on the bench's synthetic data (closed filter); no correctness check."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dhr_amd import _lib, synth
from dhr_amd.retrieval.gip_retrieval import GipIndex
import bench
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
variants = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "3,4").split(",")]
dev = torch.device("cuda", 0)
cv, ci = bench.gen_shard(torch, synth, dev, 4242, rows, 768, 768, 30, 90, False)
qv, qi = bench.gen_shard(torch, synth, dev, 777, 6980, 768, 768, 4, 12, False)
ix = GipIndex(cv, ci, idx_buckets=2)
del cv
qb, keep = _lib.make_query_batch(qv, qi)
for rep in range(2):
    for variant in variants:
        ix.set_param(_lib.PARAM_GEMM_VARIANT, variant)
        ms, fl = C.c_double(), C.c_double()
        _lib.check(ix._lib.dhr_debug_gemm_time(ix._h, C.byref(qb), 8, C.byref(ms), C.byref(fl), None), "gemm_time")
        alg = 2.0 * rows * 6980 * 1536
        print("%s variant %d: %.3f ms per %d rows, algorithmic %.1f TFLOP/s (frac %.3f)" % (os.environ.get("DHR_HIP_LIB", "default")[-8:], variant, ms.value, rows, alg / ms.value / 1e9, alg / ms.value / 1e9 / 2500), flush=True)
ix.close()
