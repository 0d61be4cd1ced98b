#!/usr/bin/env python
"""Time the shard reduce on one GPU: sorted-lists rank merge (dhr_merge_topk_lists) against the general
bitonic reduce (dhr_merge_topk) on the shapes the 8-shard step produces."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from dhr_amd import dist as D, _lib


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    q, k = 6980, 1000
    lib = _lib.load()
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]          # e.g. 8x448: only these shapes
    for n_lists, ll in shapes or [(8, 192), (8, 256), (8, 384), (8, 448), (8, 512), (4, 384), (2, 640), (8, 1000), (4, 1000), (2, 1000), (8, 58), (8, 99)]:
        s = torch.randn(n_lists, q, ll, device="cuda").sort(dim=2, descending=True).values
        r = torch.randint(0, 8_000_000, (n_lists, q, ll), device="cuda", dtype=torch.int64)
        cs = s.permute(1, 0, 2).reshape(q, -1).contiguous(); cr = r.permute(1, 0, 2).reshape(q, -1).contiguous()
        kk = min(k, n_lists * ll)
        os_ = torch.empty((q, kk), dtype=torch.float32, device="cuda"); or_ = torch.empty((q, kk), dtype=torch.int64, device="cuda")
        def direct():                                      # the kernel itself (the wrapper sends long lists to the general reduce)
            _lib.check(lib.dhr_merge_topk_lists(0, q, n_lists, ll, s.data_ptr(), r.data_ptr(), kk, os_.data_ptr(), or_.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), "merge lists")
        t_new = timeit(direct)
        t_old = timeit(lambda: D.merge_topk(cs, cr, kk))
        t_perm = timeit(lambda: (s.permute(1, 0, 2).reshape(q, -1).contiguous(), r.permute(1, 0, 2).reshape(q, -1).contiguous()))
        t_sc = timeit(lambda: D.merge_sorted_lists(s, None, min(ll, kk)))
        t_tk = timeit(lambda: torch.topk(cs, min(ll, kk), dim=1))
        print("lists %d x %4d : rank merge %.3f ms | transpose %.3f + bitonic %.3f ms | scores only %.3f ms vs torch.topk %.3f ms"
              % (n_lists, ll, t_new, t_perm, t_old, t_sc, t_tk))


if __name__ == "__main__":
    main()
