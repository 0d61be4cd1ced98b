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
    trp = getattr(lib, "dhr_debug_g8p_trace", None)
    if tr is None or trp is None:
        raise SystemExit("this library was not built with -DG8_TRACE=1")
    tr.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint)]
    trp.argtypes = [C.c_void_p, C.c_int]
    qb, keep = _lib.make_query_batch(qv, qi)
    ms, fl = C.c_double(), C.c_double()
    n = C.c_uint()
    torch.cuda.synchronize()
    cap = 1 << 18
    persistent = os.environ.get("DHR_G8_PERSIST", "1") != "0"
    if persistent:
        # persistent workgroups (gemm_g8p.hip): one record per tile at slot 1024 x workgroup + the workgroup's tile count; the warm-up launch
        # of dhr_debug_gemm_time writes the same slots first, the timed launch overwrites them.  (The trace build of THAT kernel spills 131
        # registers: its phase times are not the product's; the hit-stack statistics of tools/g8p_stat.py are unaffected.)
        assert trp(None, 0) == 0
        _lib.check(lib.dhr_debug_gemm_time(ix._h, C.byref(qb), 1, C.byref(ms), C.byref(fl), None), "gemm_time")
        torch.cuda.synchronize()
        buf = np.zeros((cap, 8), dtype=np.uint64)
        assert trp(buf.ctypes.data, cap) == 0
        r = buf[buf[:, 1] != 0].astype(np.int64)
        print("launch %.3f ms, %d tile records of the timed launch (persistent workgroups)" % (ms.value, len(r)))
    else:
        assert tr(None, 0, C.byref(n)) == 0              # drop the records of the search above
        _lib.check(lib.dhr_debug_gemm_time(ix._h, C.byref(qb), 1, C.byref(ms), C.byref(fl), None), "gemm_time")
        torch.cuda.synchronize()
        buf = np.zeros((cap, 8), dtype=np.uint64)
        assert tr(buf.ctypes.data, cap, C.byref(n)) == 0
        m = min(n.value, cap)
        print("launch %.3f ms, %d records (2 launches: warm-up + timed)" % (ms.value, n.value))
        r = buf[:m].astype(np.int64)
        r = r[np.argsort(r[:, 0])][-(len(r) // 2):]          # the timed launch
    if a.save:
        np.save(a.save, r)
    analyse(r)
    ix.close()


def analyse(r):
    """r: [records][8] = s_memrealtime at entry (100 MHz), s_memtime at entry / first pair landed / end of the gated stages / end of the
    ungated stages / accumulators final / exit, HW_ID | XCC_ID << 32 | qt << 40 | dt << 48.  s_memtime counts shader clocks with a
    base of its own on every XCD: only differences inside a workgroup or a CU mean anything."""
    r = r[np.argsort(r[:, 0])]
    rt0, t1, t2, t3, t4, t5, t6, meta = [r[:, i] for i in range(8)]
    cu = ((meta >> 32) & 0xf) * 4096 + ((meta >> 8) & 0xff)
    rates, gaps, spans = [], [], []
    for c in np.unique(cu):
        i = np.where(cu == c)[0]
        i = i[np.argsort(t1[i])]
        if len(i) < 3:
            continue
        rates.append((t1[i][-1] - t1[i][0]) / ((rt0[i][-1] - rt0[i][0]) / 100.0))
        gaps.append(t1[i][1:] - t6[i][:-1])
        spans.append(((t6[i] - t1[i]).sum(), t6[i][-1] - t1[i][0], len(i)))
    mhz = float(np.median(rates))
    print("shader clock %.0f MHz; launch spans %.3f ms; %d CUs, %.1f workgroups per CU" % (mhz, (rt0.max() - rt0.min()) / 1e5, len(spans), np.mean([x[2] for x in spans])))
    names = ["prologue (entry -> first pair landed)", "gated stages", "ungated stages", "last wait", "epilogue", "whole tile"]
    segs = [t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t6 - t1]
    g = np.concatenate(gaps)
    names.append("gap to the CU's next workgroup")
    segs.append(g)
    for nm, s in zip(names, segs):
        print("  %-40s mean %7.0f cycles = %6.3f us   median %7.0f  p10 %7.0f  p90 %7.0f" % (nm, s.mean(), s.mean() / mhz, np.median(s), np.percentile(s, 10), np.percentile(s, 90)))
    b = np.array([x[0] / x[1] for x in spans])
    print("  fraction of a CU's span inside some tile: mean %.3f min %.3f" % (b.mean(), b.min()))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--load":
        r_ = np.load(sys.argv[2])
        analyse(r_ if len(sys.argv) > 3 else r_[np.argsort(r_[:, 0])][-(len(r_) // 2):])
    else:
        main()
