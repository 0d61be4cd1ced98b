#!/usr/bin/env python
"""Time dhr_index_save / dhr_index_load against a rebuild from arrays (synthetic hybrid shard)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2_000_000)
    ap.add_argument("--path", default="/tmp/dhr_index_file_bench.dhr")
    a = ap.parse_args()
    import torch
    import bench
    from dhr_amd import synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    dev = torch.device("cuda", 0)
    cv, ci = bench.gen_shard(torch, synth, dev, 4242, a.rows, 768, 768, 30, 90, False)
    qv, qi = bench.gen_shard(torch, synth, dev, 777, 256, 768, 768, 4, 12, False)
    torch.cuda.synchronize(); t = time.perf_counter()
    ix = GipIndex(cv, ci)
    torch.cuda.synchronize(); t_build_dev = time.perf_counter() - t
    s0, r0 = ix.search(qv, qi, 100)
    hv, hi = cv.cpu().numpy(), ci.cpu().numpy()
    del cv, ci
    t = time.perf_counter(); ix.save(a.path, None); t_save = time.perf_counter() - t
    ix.close()
    size = os.path.getsize(a.path)
    t = time.perf_counter(); ixh = GipIndex(hv, hi); t_build_host = time.perf_counter() - t
    ixh.close()
    t = time.perf_counter(); ix2, _ = GipIndex.load(a.path); t_load = time.perf_counter() - t
    s1, r1 = ix2.search(qv, qi, 100)
    ix2.close()
    os.unlink(a.path)
    same = bool((r0 == r1).all() and (s0 == s1).all())
    print("rows %d: file %.2f GB; save %.2f s (%.2f GB/s); load (page cache warm) %.2f s (%.2f GB/s); build from host arrays %.2f s; "
          "build from device arrays %.2f s; results identical: %s"
          % (a.rows, size / 1e9, t_save, size / 1e9 / t_save, t_load, size / 1e9 / t_load, t_build_host, t_build_dev, same))


if __name__ == "__main__":
    main()
