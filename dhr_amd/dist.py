"""Row-sharded retrieval across the GPUs of one node: one process per GPU, the corpus split exactly
like the reference's --total_shrad/--shrad (retrieval/gip_retrieval.py:292-306), queries replicated,
all-gathers of the per-shard results over RCCL (xGMI) and a per-query reduce on every rank -- the
semantics of retrieval/merge.result.py:22-42 without the text-file round trip.

The product path is the C ABI: `ShardComm` (dhr_comm_*: an RCCL communicator the library creates itself
from a 128-byte id; the id travels over whatever torch.distributed group is up) and
`sharded_search` -> dhr_search_sharded (sample all-gather, common thresholds, count all-gather, list
all-gather, rank merge: dhr_amd/csrc/sharded.hip).  The same control flow written with torch.distributed
collectives (`sharded_search_torch`) is kept as the CPU test shim: it runs under gloo, where RCCL cannot
(tests/test_dist_gloo.py), and is what `sharded_search` uses when the process group's backend is gloo."""
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


def merge_sorted_lists(scores, rows, k: int):
    """scores/rows: [n_lists, Q, L] tensors, every list sorted (score desc, row asc), padding (row < 0) at
    its tail -- the layout all_gather_into_tensor leaves the per-shard results in.  -> ([Q,k], [Q,k]).
    rows may be None (scores only; returns (scores, None)).  No sort, no transpose copy: dhr_merge_topk_lists."""
    import torch
    lib = _lib.load()
    n_lists, q, ll = (int(x) for x in scores.shape)
    scores = scores.contiguous()
    rows = rows.contiguous() if rows is not None else None
    if n_lists > 64 or (n_lists * ll + k) * (12 if rows is not None else 4) > 160 * 1024:   # beyond the LDS: general reduce
        cs = scores.permute(1, 0, 2).reshape(q, n_lists * ll)
        cr = rows.permute(1, 0, 2).reshape(q, n_lists * ll) if rows is not None else \
            torch.arange(n_lists * ll, device=scores.device, dtype=torch.int64).expand(q, -1).contiguous()
        out = merge_topk(cs, cr, k)
        return out if rows is not None else (out[0], None)
    out_s = torch.empty((q, k), dtype=torch.float32, device=scores.device)
    out_r = torch.empty((q, k), dtype=torch.int64, device=scores.device) if rows is not None else None
    pr = rows.data_ptr() if rows is not None else None
    po = out_r.data_ptr() if rows is not None else None
    if scores.is_cuda:
        stream = torch.cuda.current_stream(scores.device).cuda_stream
        _lib.check(lib.dhr_merge_topk_lists(scores.device.index, q, n_lists, ll, scores.data_ptr(), pr, k,
                                            out_s.data_ptr(), po, stream), "dhr_merge_topk_lists")
    else:
        _lib.check(lib.dhr_merge_topk_lists_host(q, n_lists, ll, scores.data_ptr(), pr, k, out_s.data_ptr(), po),
                   "dhr_merge_topk_lists_host")
    return out_s, out_r


def allgather_merge(local_scores, local_rows, k: int, group=None):
    """local_* : [Q, k_local] tensors of this rank's shard (global rows, sorted best first as the search
    returns them).  Returns the merged [Q, k] lists, identical on every rank."""
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
    # [world, Q, kl], shards in rank order (ascending row ranges): reduced in place, no transpose
    return merge_sorted_lists(gs.view(world, q, kl), gr.view(world, q, kl), k)


def common_threshold(sample_scores_all, r: int):
    """[world, Q, r_local] best sample scores of every shard (each list sorted, best first) -> [Q] the r-th best of the union per
    query (r = GipIndex.union_rank; the lists may be shorter than r when DHR_PARAM_SAMPLE_SHARE is set: then the
    min(r, world * r_local)-th best, which can only be lower)."""
    r = min(int(r), sample_scores_all.shape[0] * sample_scores_all.shape[2])
    merged, _ = merge_sorted_lists(sample_scores_all, None, r)
    return merged[:, r - 1].contiguous()


class ShardComm:
    """dhr_comm: the library's own RCCL communicator for this rank.  Collective constructor: rank 0 draws the id
    (dhr_comm_unique_id), it is broadcast over the torch.distributed group, every rank calls dhr_comm_create."""

    def __init__(self, device: int, group=None):
        import torch
        import torch.distributed as dist
        self._lib = _lib.load()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = int(device)
        uid = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(self._lib.dhr_comm_unique_id(uid, 128), "dhr_comm_unique_id")
        if self.world > 1:
            backend = dist.get_backend(group)
            t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
            if backend == "nccl":
                t = t.to(torch.device("cuda", self.device))
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
        h = C.c_void_p()
        _lib.check(self._lib.dhr_comm_create(uid, self.world, self.rank, self.device, C.byref(h)), "dhr_comm_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dhr_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_COMMS = {}


def _comm_for(index, group):
    key = (index.device, id(group))
    if key not in _COMMS:
        _COMMS[key] = ShardComm(index.device, group)
    return _COMMS[key]


def search_sharded_local(shards, q_value, q_index, k: int):
    """One process, several shard handles (dhr_search_sharded_local): -> (scores, rows) torch cuda tensors [Q, k]."""
    import torch
    lib = _lib.load()
    qb, keep = shards[0]._qb(q_value, q_index)
    dev = torch.device("cuda", shards[0].device)
    scores = torch.empty((qb.n_queries, k), dtype=torch.float32, device=dev)
    rows = torch.empty((qb.n_queries, k), dtype=torch.int64, device=dev)
    arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.dhr_search_sharded_local(arr, len(shards), C.byref(qb), int(k), scores.data_ptr(), rows.data_ptr(), _lib.MEM_DEVICE, stream),
                   "dhr_search_sharded_local")
    return scores, rows


def sharded_search(index, q_value, q_index, k: int, group=None):
    """index: this rank's GipIndex (rows shard_bounds(N, world, rank), row_offset=lo).  Returns the global [Q,k] (scores, rows)
    torch cuda tensors on every rank.  RCCL process groups (and single processes) run dhr_search_sharded; a gloo group runs the
    torch.distributed restatement of the same steps (testing)."""
    import torch
    import torch.distributed as dist
    import os
    if dist.is_initialized() and dist.get_world_size(group) > 1 and (dist.get_backend(group) != "nccl" or os.environ.get("DHR_SHARDED_IMPL") == "torch"):
        return sharded_search_torch(index, q_value, q_index, k, group)      # gloo (CPU tests), or forced for A/B debugging on RCCL
    comm = _comm_for(index, group)
    lib = _lib.load()
    qb, keep = index._qb(q_value, q_index)
    dev = torch.device("cuda", index.device)
    scores = torch.empty((qb.n_queries, k), dtype=torch.float32, device=dev)
    rows = torch.empty((qb.n_queries, k), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.dhr_search_sharded(index._h, comm._h, C.byref(qb), int(k), scores.data_ptr(), rows.data_ptr(), _lib.MEM_DEVICE, stream),
                   "dhr_search_sharded")
    return scores, rows


def sharded_search_torch(index, q_value, q_index, k: int, group=None):
    """index: this rank's GipIndex (rows shard_bounds(N, world, rank), row_offset=lo).  Returns the global
    [Q,k] (scores, rows) on every rank.

    The shards agree on ONE threshold per query after their sampled runs (all-gather of [Q, r] scores,
    r ~ 100), so a shard collects only its share of the global top-k instead of a full local top-k;
    completeness is verified with one all-reduce of per-query counts, and the queries that fail it
    (unrepresentative sample, list overflow) are redone with purely local thresholds."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        scores, rows = index.search(q_value, q_index, k, out_device=True)
        return merge_topk(scores, rows, k)
    index.set_param(_lib.PARAM_SAMPLE_SHARE, world)                   # a shard reports only its plausible share of the union's r best
    r = index.sample_rank(k)
    dev = getattr(index, "torch_device", None) or torch.device("cuda", index.device)
    rr = torch.tensor([r, -r], dtype=torch.int32, device=dev)
    dist.all_reduce(rr, op=dist.ReduceOp.MIN, group=group)            # (min r, -max r): shards of different size may disagree
    rr = rr.tolist()
    if r == 0 or rr[0] != r or -rr[1] != r:                           # not uniformly samplable: local thresholds
        scores, rows = index.search(q_value, q_index, k, out_device=True)
        return allgather_merge(scores, rows, k, group)
    sample = index.search_begin(q_value, q_index, k)
    nq = sample.shape[0]
    gathered = torch.empty((world * nq, r), dtype=torch.float32, device=sample.device)
    dist.all_gather_into_tensor(gathered, sample, group=group)
    tau = common_threshold(gathered.view(world, nq, r), index.union_rank(k))
    scores, rows, count = index.search_finish(tau)
    # one small all-gather of the per-query counts serves the completeness check AND the useful list length
    counts = torch.empty((world * nq,), dtype=torch.int32, device=count.device)
    dist.all_gather_into_tensor(counts, count.to(torch.int32).contiguous(), group=group)
    counts = counts.view(world, nq)
    fail_mask = (counts.clamp(min=0).sum(0) < k) | (counts < 0).any(0)
    n_failed, cmax = torch.stack([fail_mask.sum(), counts.max()]).tolist()      # one host read for both
    failed = torch.nonzero(fail_mask).flatten() if n_failed else fail_mask[:0]
    if failed.numel() > 0:                                             # identical on every rank
        ids = failed.cpu().numpy()
        sub_v = q_value[ids] if not hasattr(q_value, "index_select") else q_value.index_select(0, failed.to(q_value.device))
        sub_i = None
        if q_index is not None:
            sub_i = q_index[ids] if not hasattr(q_index, "index_select") else q_index.index_select(0, failed.to(q_index.device))
        fs, fr = index.search(sub_v, sub_i, k, out_device=True)
        scores[failed] = fs
        rows[failed] = fr
        return allgather_merge(scores, rows, k, group)
    # every global top-k row reaches tau, and a shard holds `count` of those: the tails of the lists are dead
    kk = min(k, (int(cmax) + 63) // 64 * 64)
    return allgather_merge(scores[:, :kk].contiguous(), rows[:, :kk].contiguous(), k, group)
