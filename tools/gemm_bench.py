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
    ap.add_argument("--variant", type=int, default=5)
    ap.add_argument("--dlr", type=int, default=0, help="gated columns (with a random uint8 slice index); the rest of --k is dense")
    ap.add_argument("--idx-buckets", type=int, default=0)
    ap.add_argument("--synth", action="store_true", help="the bench's synthetic hybrid data instead of uniform random operands")
    ap.add_argument("--open", action="store_true", help="run a search first and keep its final thresholds in the filter (DHR_GEMM_TIME_OPEN)")
    a = ap.parse_args()
    import torch
    from dhr_amd import _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    g = torch.Generator(device="cuda").manual_seed(1)
    cv = (torch.randn((a.rows, a.k), generator=g, device="cuda") * 0.1).half()
    qv = (torch.randn((a.queries, a.k), generator=g, device="cuda") * 0.1).half()
    ci = qi = None
    if a.dlr:
        cv[:, :a.dlr] = cv[:, :a.dlr].abs(); qv[:, :a.dlr] = qv[:, :a.dlr].abs()
        ci = torch.randint(0, 39, (a.rows, a.dlr), generator=g, device="cuda", dtype=torch.uint8)
        qi = torch.randint(0, 39, (a.queries, a.dlr), generator=g, device="cuda", dtype=torch.uint8)
    if a.synth:
        import bench
        from dhr_amd import synth
        dev = torch.device("cuda", 0)
        cv, ci = bench.gen_shard(torch, synth, dev, 4242, a.rows, a.dlr or 768, a.k - (a.dlr or 768), 30, 90, False)
        qv, qi = bench.gen_shard(torch, synth, dev, 777, a.queries, a.dlr or 768, a.k - (a.dlr or 768), 4, 12, False)
    ix = GipIndex(cv, ci, idx_buckets=a.idx_buckets)
    if a.open:
        os.environ["DHR_GEMM_TIME_OPEN"] = "1"
        ix.search(qv, qi, 1000, out_device=True)
    del cv
    if a.variant in (4, 5):          # (6: A/B builds only, tools/ab)
        ix.set_param(_lib.PARAM_GEMM_VARIANT, a.variant)
    qb, keep = _lib.make_query_batch(qv, qi)
    ms, fl = C.c_double(), C.c_double()
    _lib.check(ix._lib.dhr_debug_gemm_time(ix._h, C.byref(qb), a.iters, C.byref(ms), C.byref(fl), None), "gemm_time")
    print("variant %d dlr %d buckets %d rows %d k %d q %d : %.3f ms/launch  %.1f TFLOP/s issued" % (a.variant, a.dlr, a.idx_buckets, a.rows, a.k, a.queries, ms.value, fl.value / ms.value / 1e9))
    ix.close()


if __name__ == "__main__":
    main()
