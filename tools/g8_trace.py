#!/usr/bin/env python
"""Per-tile timeline of gemm_filter_g8_kernel from a -DG8_TRACE=1 build of the library (tools/ab_build.sh trace "-DG8_TRACE=1";
DHR_HIP_LIB=dhr_amd/csrc/_ab/libdhr_hip_trace.so): thread 0 of every workgroup records s_memtime at the tile's phase boundaries and its
CU; this prints where a tile's time goes and how long a CU waits between two workgroups.  Timing tool, not a test."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=500_000)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--dlr", type=int, default=768)
    ap.add_argument("--cls", type=int, default=768)
    ap.add_argument("--open", action="store_true")
    ap.add_argument("--save", default="")
    a = ap.parse_args()
    import torch
    import bench
    from dhr_amd import _lib, synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    cv, ci = bench.gen_shard(torch, synth, dev, 4242, a.rows, a.dlr, a.cls, 30, 90, False)
    qv, qi = bench.gen_shard(torch, synth, dev, 777, a.queries, a.dlr, a.cls, 4, 12, False)
    ix = GipIndex(cv, ci)
    if a.open:
        os.environ["DHR_GEMM_TIME_OPEN"] = "1"
        ix.search(qv, qi, 1000, out_device=True)
    del cv
    lib = ix._lib
    tr = getattr(lib, "dhr_debug_g8_trace", None)
    if tr is None:
        raise SystemExit("this library was not built with -DG8_TRACE=1")
    tr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint)]
    qb, keep = _lib.make_query_batch(qv, qi)
    ms, fl = C.c_double(), C.c_double()
    n = C.c_uint()
    _lib.check(lib.dhr_debug_gemm_time(ix._h, C.byref(qb), 1, C.byref(ms), C.byref(fl), None), "gemm_time")
    torch.cuda.synchronize()
    cap = 1 << 18
    buf = np.zeros((cap, 8), dtype=np.uint64)
    assert tr(buf.ctypes.data, cap, C.byref(n)) == 0
    m = min(n.value, cap)
    print("launch %.3f ms, %d records (2 launches: warm-up + timed)" % (ms.value, n.value))
    r = buf[:m].astype(np.int64)
    if a.save:
        np.save(a.save, r)
    analyse(r)
    ix.close()


def analyse(r):
    half = len(r) // 2
    r = r[half:]                                   # the timed launch
    rt0, t1, t2, t3, t4, t5, t6, meta = [r[:, i] for i in range(8)]
    # s_memtime ticks per microsecond (s_memrealtime counts 100 MHz)
    o = np.argsort(t1)
    tick_us = (t1[o[-1]] - t1[o[0]]) / max(1, (rt0[o[-1]] - rt0[o[0]])) * 100.0
    print("s_memtime ticks per us: %.2f; launch spans %.3f ms" % (tick_us, (t6.max() - t1.min()) / tick_us / 1e3))
    ok = (t2 > 0) & (t6 > 0)
    def us(x):
        return x / tick_us
    names = ["prologue (entry -> first pair landed)", "gated stages", "ungated stages", "last wait", "epilogue", "whole tile"]
    segs = [t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t1]
    for nm, s in zip(names, segs):
        s = us(s[ok])
        print("  %-40s mean %7.3f us  median %7.3f  p10 %7.3f  p90 %7.3f" % (nm, s.mean(), np.median(s), np.percentile(s, 10), np.percentile(s, 90)))
    cu = (meta & 0xffffffffff) >> 8                # cu / sh / se bits of HW_ID + the XCC id
    cu = ((meta >> 32) & 0xf) * 4096 + ((meta >> 8) & 0xff)
    gaps = []
    busy = []
    for c in np.unique(cu):
        i = np.where(cu == c)[0]
        i = i[np.argsort(t1[i])]
        g = t1[i][1:] - t6[i][:-1]
        gaps.append(g)
        busy.append(((t6[i] - t1[i]).sum(), t6[i].max() - t1[i].min(), len(i)))
    g = us(np.concatenate(gaps))
    print("  CUs seen %d; workgroups per CU %.1f" % (len(busy), np.mean([b[2] for b in busy])))
    print("  gap between two workgroups on a CU: mean %.3f us  median %.3f  p10 %.3f  p90 %.3f  (negative = two workgroups overlapped)" % (g.mean(), np.median(g), np.percentile(g, 10), np.percentile(g, 90)))
    b = np.array([(x[0] / x[1]) for x in busy])
    print("  fraction of a CU's span inside some tile: mean %.3f min %.3f" % (b.mean(), b.min()))
    sp = us(np.array([x[1] for x in busy]))
    print("  CU span: mean %.1f us min %.1f max %.1f" % (sp.mean(), sp.min(), sp.max()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--load":
        analyse(np.load(sys.argv[2]))
    else:
        main()
