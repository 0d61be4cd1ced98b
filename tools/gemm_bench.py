#!/usr/bin/env python
"""Time the bound GEMM kernel alone (filter closed): random fp16 operands, N rows x K columns dense-only
index, Q queries.  Prints issued TFLOP/s.  Used for kernel tuning and for PMC runs."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--k", type=int, default=1536)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--variant", type=int, default=1)
    a = ap.parse_args()
    import torch
    from dhr_amd import _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    g = torch.Generator(device="cuda").manual_seed(1)
    cv = (torch.randn((a.rows, a.k), generator=g, device="cuda") * 0.1).half()
    qv = (torch.randn((a.queries, a.k), generator=g, device="cuda") * 0.1).half()
    ix = GipIndex(cv, None)
    del cv
    ix.set_param(_lib.PARAM_GEMM_VARIANT, a.variant)
    qb, keep = _lib.make_query_batch(qv, None)
    ms, fl = C.c_double(), C.c_double()
    _lib.check(ix._lib.dhr_debug_gemm_time(ix._h, C.byref(qb), a.iters, C.byref(ms), C.byref(fl), None), "gemm_time")
    print("variant %d rows %d k %d q %d : %.3f ms/launch  %.1f TFLOP/s issued" % (a.variant, a.rows, a.k, a.queries, ms.value, fl.value / ms.value / 1e9))
    ix.close()


if __name__ == "__main__":
    main()
