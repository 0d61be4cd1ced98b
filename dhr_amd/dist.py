"""Row-sharded retrieval across the GPUs of one node: one process per GPU, the corpus split exactly
like the reference's --total_shrad/--shrad (retrieval/gip_retrieval.py:292-306), queries replicated,
ONE all-gather of the per-shard top-k (RCCL over xGMI; `nccl` backend) and a per-query k-way reduce
on every rank -- the semantics of retrieval/merge.result.py:22-42 without the text-file round trip.

Payload per rank at Q=6980, k=1000: 6980*1000*(4+8) B = 84 MB."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def shard_bounds(n_docs: int, world: int, rank: int):
    """gip_retrieval.py:292-306: per = n // world; the last shard takes the remainder."""
    per = n_docs // world
    lo = per * rank
    hi = n_docs if rank == world - 1 else per * (rank + 1)
    return lo, hi


def merge_topk(scores, rows, k: int):
    """Per query the k best of the concatenated (score,row) lists, (score desc, row asc); row<0 is
    padding.  torch tensors [Q, n_in]; CUDA tensors use the device kernel, CPU tensors the host twin."""
    import torch
    lib = _lib.load()
    q, n_in = int(scores.shape[0]), int(scores.shape[1])
    scores = scores.contiguous()
    rows = rows.contiguous()
    out_s = torch.empty((q, k), dtype=torch.float32, device=scores.device)
    out_r = torch.empty((q, k), dtype=torch.int64, device=scores.device)
    if scores.is_cuda:
        stream = torch.cuda.current_stream(scores.device).cuda_stream
        _lib.check(lib.dhr_merge_topk(scores.device.index, q, n_in, scores.data_ptr(), rows.data_ptr(), k,
                                      out_s.data_ptr(), out_r.data_ptr(), stream), "dhr_merge_topk")
    else:
        _lib.check(lib.dhr_merge_topk_host(q, n_in, scores.data_ptr(), rows.data_ptr(), k, out_s.data_ptr(),
                                           out_r.data_ptr()), "dhr_merge_topk_host")
    return out_s, out_r


def allgather_merge(local_scores, local_rows, k: int, group=None):
    """local_* : [Q, k_local] tensors of this rank's shard (global rows).  Returns the merged
    [Q, k] lists, identical on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return merge_topk(local_scores, local_rows, k)
    q, kl = local_scores.shape
    gs = torch.empty((world * q, kl), dtype=torch.float32, device=local_scores.device)
    gr = torch.empty((world * q, kl), dtype=torch.int64, device=local_rows.device)
    dist.all_gather_into_tensor(gs, local_scores.contiguous(), group=group)
    dist.all_gather_into_tensor(gr, local_rows.contiguous(), group=group)
    # [world, Q, kl] -> [Q, world*kl], shards in rank order (ascending row ranges)
    cs = gs.view(world, q, kl).permute(1, 0, 2).reshape(q, world * kl)
    cr = gr.view(world, q, kl).permute(1, 0, 2).reshape(q, world * kl)
    return merge_topk(cs, cr, k)


def sharded_search(index, q_value, q_index, k: int, group=None):
    """index: this rank's GipIndex (built on rows shard_bounds(N, world, rank) with row_offset=lo)."""
    scores, rows = index.search(q_value, q_index, k, out_device=True)
    return allgather_merge(scores, rows, k, group)
