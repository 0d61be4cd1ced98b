"""Row-sharded retrieval across the GPUs of one node: one process per GPU, the corpus split exactly
like the reference's --total_shrad/--shrad (retrieval/gip_retrieval.py:292-306), queries replicated,
all-gathers of the per-shard results over RCCL (xGMI) and a per-query reduce on every rank -- the
semantics of retrieval/merge.result.py:22-42 without the text-file round trip.

The product path is the C ABI: `ShardComm` (dhr_comm_*: an RCCL communicator the library creates itself
from a 128-byte id; the id travels over whatever torch.distributed group is up) and
`sharded_search` -> dhr_search_sharded (sample all-gather, common thresholds, count all-gather, list
all-gather, rank merge: dhr_amd/csrc/sharded.hip).  There is ONE implementation of that control flow, in the
library: where RCCL is not available between the ranks (a gloo group: several ranks on one GPU, CPU tests)
the all-gathers are handed to torch.distributed through a callback communicator, and the CPU test-suite drives
the same control flow over oracle-backed host shards (`sharded_search_host` -> dhr_search_sharded_host).  (Until
round 4 a torch restatement of the steps lived here and was what the gloo tests exercised.)"""
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


def _host_allgather(group, device=None):
    """dhr_allgather_fn over a torch.distributed group: `bytes` from every rank -> recv = [world][bytes] (host buffers).  gloo gathers the
    host bytes as they are; an RCCL group (which only moves device tensors) stages them through `device` (the communicator's)."""
    import torch
    import torch.distributed as dist

    def fn(_user, send, recv, nbytes):
        try:
            world = dist.get_world_size(group)
            src = torch.frombuffer((C.c_char * nbytes).from_address(send), dtype=torch.uint8)
            dst = torch.frombuffer((C.c_char * (nbytes * world)).from_address(recv), dtype=torch.uint8)
            if dist.get_backend(group) == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
                out = torch.empty(nbytes * world, dtype=torch.uint8, device=dev)
                dist.all_gather_into_tensor(out, src.to(dev), group=group)
                dst.copy_(out.cpu())
            else:
                dist.all_gather_into_tensor(dst, src.clone(), group=group)
            return 0
        except Exception as e:  # noqa: BLE001
            import sys
            print("[dhr] all-gather callback failed: %r" % (e,), file=sys.stderr)
            return 1
    return _lib.ALLGATHER_FN(fn)


class ShardComm:
    """dhr_comm for this rank.  Collective constructor.  transport "rccl": the library's own RCCL communicator (rank 0 draws the id --
    dhr_comm_unique_id --, it is broadcast over the torch.distributed group, every rank calls dhr_comm_create); transport "host": the
    all-gathers are done by torch.distributed on host buffers (dhr_comm_create_callback) -- what a gloo group uses (several ranks on
    one GPU, CPU-only process groups).  Either way the search is the library's one control flow (sharded.hip sharded_core).

    This constructor does NOT fall back: a dhr_comm_create that fails raises, and one that never returns (a peer that did not reach
    ncclCommInitRank) blocks.  `bring_up` below is the constructor with a watchdog and a collective fallback to the host transport.
    The host transport is a bring-up / test transport: a step has 5-7 gathers, each a device -> host copy, a stream synchronisation
    and a blocking callback, and a failure on ONE rank between two gathers leaves the others waiting in the next one until the
    torch.distributed group's own timeout fires (give the group a short timeout: bring_up's control group has one)."""

    def __init__(self, device: int, group=None, transport: str | None = None, uid: bytes | None = None):
        import torch
        import torch.distributed as dist
        self._lib = _lib.load()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = int(device)
        self.note = ""
        if transport is None:
            transport = "rccl" if (self.world == 1 or dist.get_backend(group) == "nccl") else "host"
        self.transport = transport
        h = C.c_void_p()
        if transport == "host":
            self._cb = _host_allgather(group, self.device)           # keep the ctypes thunk alive as long as the communicator
            _lib.check(self._lib.dhr_comm_create_callback(self.world, self.rank, self.device, self._cb, None, C.byref(h)), "dhr_comm_create_callback")
            self._h = h
            return
        if uid is None:
            buf = (C.c_char * 128)()
            if self.rank == 0:
                _lib.check(self._lib.dhr_comm_unique_id(buf, 128), "dhr_comm_unique_id")
            if self.world > 1:
                backend = dist.get_backend(group)
                t = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone()
                if backend == "nccl":
                    t = t.to(torch.device("cuda", self.device))
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                uid = bytes(t.cpu().numpy().tobytes())
            else:
                uid = bytes(buf)
        _lib.check(self._lib.dhr_comm_create((C.c_char * 128).from_buffer_copy(uid), self.world, self.rank, self.device, C.byref(h)), "dhr_comm_create")
        self._h = h

    def info(self, what: int) -> int:
        """dhr_comm_info: what the transport itself reports (RCCL: ncclCommCount / ncclCommUserRank / ncclCommCuDevice)."""
        v = int(self._lib.dhr_comm_info(self._h, int(what)))
        if v < 0:
            _lib.check(v, "dhr_comm_info")
        return v

    def ranks_seen(self) -> int:
        return self.info(_lib.COMM_WORLD)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dhr_comm_destroy(self._h)
            self._h = None

    def abort(self):
        """ncclCommAbort: callable while another thread of this process is blocked in one of the communicator's collectives.  The handle
        is only marked dead (the blocked thread unwinds through code that still uses it); close() frees it -- call that once the thread
        is back.  A handle that is never closed after an abort is a small leak, never a use-after-free."""
        if getattr(self, "_h", None):
            self._lib.dhr_comm_abort(self._h)
            self._aborted = True

    def __del__(self):
        try:
            if not getattr(self, "_aborted", False):      # (an aborted handle may still be in use by the thread it unblocked: bring_up closes it)
                self.close()
        except Exception:  # noqa: BLE001
            pass


_COMMS = {}
_CTL_GROUPS = {}


def control_group(group=None, timeout_s: float = 120.0):
    """A gloo group over the same ranks, with a timeout: votes and the host transport run on it, so that nothing the bring-up of RCCL
    does (or fails to do) on one rank can leave the others waiting forever.  Collective; cached per parent group."""
    import datetime
    import torch.distributed as dist
    if dist.get_backend(group) == "gloo":
        return group
    key = id(group)
    if key not in _CTL_GROUPS:
        ranks = dist.get_process_group_ranks(group) if group is not None else None
        _CTL_GROUPS[key] = dist.new_group(ranks=ranks, backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))
    return _CTL_GROUPS[key]


def bring_up(device: int, group=None, trial=None, timeout_s: float = 120.0, want: str = "rccl"):
    """ShardComm with a watchdog and a COLLECTIVE fallback.  Every rank calls it; every rank returns a communicator of the same
    transport.  RCCL is brought up (id from rank 0 over the control group, dhr_comm_create, then `trial(comm)` -- one untimed sharded
    step, synchronised) in a worker thread; the calling thread waits at most `timeout_s`, then the ranks vote over the gloo control
    group (MIN of "my bring-up finished without an exception").  One "no" -- an exception on one rank, or a rank still blocked in
    ncclCommInitRank / a collective because a peer never arrived -- and EVERY rank drops RCCL (ncclCommAbort if the handle exists; a
    thread still inside ncclCommInitRank is left behind as a daemon) and returns the host transport over the control group: the same
    control flow in the library, only the gathers change.  The calling thread itself only ever blocks in gloo collectives with a
    timeout.  `comm.note` says what happened.  (Test hooks: DHR_TEST_COMM_FAIL_RANK=r raises on rank r before dhr_comm_create,
    DHR_TEST_COMM_HANG_RANK=r blocks rank r's bring-up thread -- the shape of a peer that never reaches the collective.)"""
    import os
    import threading
    import time
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        comm = ShardComm(device, group, "rccl")
        if trial is not None:
            trial(comm)
        return comm
    ctl = control_group(group, timeout_s)
    if want != "rccl" or os.environ.get("DHR_SHARDED_TRANSPORT") == "host":
        comm = ShardComm(device, ctl, "host")
        comm.note = "host transport requested"
        if trial is not None:
            trial(comm)
        return comm
    lib = _lib.load()
    # the id travels over the control group, with a status byte: a rank 0 that cannot draw one must not leave the others in a broadcast
    t = torch.zeros(129, dtype=torch.uint8)
    if rank == 0:
        try:
            buf = (C.c_char * 128)()
            _lib.check(lib.dhr_comm_unique_id(buf, 128), "dhr_comm_unique_id")
            t[0] = 1
            t[1:] = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8)
        except Exception as e:  # noqa: BLE001
            print("[dhr] rank 0 could not draw an RCCL id: %r" % (e,))
    dist.broadcast(t, src=dist.get_global_rank(ctl, 0) if ctl is not None else 0, group=ctl)
    state = {"comm": None, "err": None, "done": False}
    if int(t[0]) == 1:
        uid = bytes(t[1:].numpy().tobytes())

        def work():
            try:
                if torch.cuda.is_available():
                    torch.cuda.set_device(device)
                if os.environ.get("DHR_TEST_COMM_FAIL_RANK", "") == str(rank):
                    raise _lib.DhrError("injected failure before dhr_comm_create (DHR_TEST_COMM_FAIL_RANK)")
                if os.environ.get("DHR_TEST_COMM_HANG_RANK", "") == str(rank):
                    time.sleep(1e6)
                state["comm"] = ShardComm(device, group, "rccl", uid=uid)
                if trial is not None:
                    trial(state["comm"])
                state["done"] = True
            except Exception as e:  # noqa: BLE001
                state["err"] = e
        th = threading.Thread(target=work, daemon=True, name="dhr-rccl-bring-up")
        th.start()
        th.join(timeout_s)
        hung = th.is_alive()
    else:
        hung = False
        state["err"] = _lib.DhrError("rank 0 could not draw an RCCL id")
    ok = state["done"] and not hung and state["err"] is None
    vote = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(vote, op=dist.ReduceOp.MIN, group=ctl)
    if int(vote.item()) == 1:
        comm = state["comm"]
        comm.note = "RCCL communicator of the library; ncclCommCount = %d" % comm.ranks_seen()
        return comm
    why = ("bring-up still blocked after %.0f s (a peer never arrived)" % timeout_s) if hung else \
        ("%s" % (state["err"],))[:200] if state["err"] is not None else "another rank failed"
    c = state["comm"]
    if c is not None:
        try:
            c.abort()              # also unblocks a worker thread parked in one of this communicator's collectives
            th.join(5.0)           # ... which unwinds through the library with the handle still in its hands: free it only once it is back
            if not th.is_alive():
                c.close()
        except Exception:  # noqa: BLE001
            pass
    comm = ShardComm(device, ctl, "host")
    comm.note = "host transport (torch.distributed gloo all-gathers on host buffers); the library's RCCL communicator was dropped on every rank: rank %d: %s" % (rank, why)
    if trial is not None:
        trial(comm)
    return comm


def _comm_for(index, group):
    import os
    transport = "host" if os.environ.get("DHR_SHARDED_TRANSPORT") == "host" else None      # force torch.distributed gathers (A/B, bring-up)
    for key in ((index.device, id(group), "checked"), (index.device, id(group), transport)):
        if key in _COMMS:
            return _COMMS[key]
    key = (index.device, id(group), transport)
    _COMMS[key] = ShardComm(index.device, group, transport)
    return _COMMS[key]


def checked_comm(index, group=None, trial=None, timeout_s: float = 120.0, want: str = "rccl"):
    """bring_up for `sharded_search`: the communicator every later sharded_search(index, ..., group) uses.  Collective."""
    key = (index.device, id(group), "checked")
    if key not in _COMMS:
        _COMMS[key] = bring_up(index.device, group, trial, timeout_s, want)
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


def sharded_search(index, q_value, q_index, k: int, group=None, comm=None):
    """index: this rank's GipIndex (rows shard_bounds(N, world, rank), row_offset=lo).  Returns the global [Q,k] (scores, rows)
    torch cuda tensors on every rank: dhr_search_sharded, over RCCL on an nccl process group and over torch.distributed host gathers
    on any other (ShardComm) -- the same control flow in the library either way."""
    import torch
    comm = comm if comm is not None else _comm_for(index, group)
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


def sharded_search_host(shard, q_value, q_index, k: int, group=None):
    """The library's sharded control flow (sharded.hip sharded_core) over a shard implemented in PYTHON on host memory -- no GPU is
    touched (dhr_search_sharded_host).  `shard` offers sample_rank(k, share), union_rank(k), search_begin(q, qi, k, share) ->
    [Q, r] float32, search_finish(tau [Q]) -> (scores [Q,k] f32, rows [Q,k] i64, count [Q] i32), search(q, qi, k) -> (scores, rows),
    all numpy; optionally mid_ranks / search_mid (second agreement) and pre_ranks(k, share) -> (r_local, r_union) / search_pre(q, qi, k,
    share, r_local) -> [Q, r_local] / search_begin_rest(tau [Q]) -> [Q, r] (first agreement in two rounds).  The all-gathers run over the torch.distributed group.  The CPU test-suite drives the shipped control flow this way."""
    import torch.distributed as dist
    lib = _lib.load()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    q_value = np.ascontiguousarray(q_value)
    q_index = None if q_index is None else np.ascontiguousarray(q_index)
    nq, width = q_value.shape
    state = {}

    def batch_arrays(qb):
        qb = qb.contents
        n = int(qb.n_queries)
        dt = np.float32 if qb.value_dtype == _lib.VAL_F32 else np.float16
        v = np.ctypeslib.as_array(C.cast(qb.value, C.POINTER(C.c_uint8)), shape=(n * int(qb.ld_value) * np.dtype(dt).itemsize,)).view(dt).reshape(n, int(qb.ld_value))[:, :width]
        x = None
        if qb.index:
            it = {_lib.IDX_U8: np.uint8, _lib.IDX_I8: np.int8, _lib.IDX_I16: np.int16}[qb.index_dtype]
            x = np.ctypeslib.as_array(C.cast(qb.index, C.POINTER(C.c_uint8)), shape=(n * int(qb.ld_index) * np.dtype(it).itemsize,)).view(it).reshape(n, int(qb.ld_index))
        return v, x

    def out(ptr, shape, dt):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(n * np.dtype(dt).itemsize,)).view(dt).reshape(shape)

    def fit(a, cols, fill, dt):
        """The callback contract is a full [n, cols] array (scores padded with -inf, rows with -1): a shard object that returns SHORTER lists (k beyond
        its rows, fewer sample scores than the agreed rank) is padded here, a longer one cut -- written as it came, a short block would be read with the
        wrong row stride (queries mixed, a stale tail merged as if it were a result)."""
        a = np.asarray(a, dt)
        if a.ndim == 1:
            a = a.reshape(-1, 1) if cols == 1 else a.reshape(1, -1)
        if a.shape[1] == cols:
            return a
        o = np.full((a.shape[0], cols), fill, dt)
        m = min(cols, a.shape[1])
        o[:, :m] = a[:, :m]
        return o

    def guard(f):
        def g(*a):
            try:
                return f(*a)
            except Exception as e:  # noqa: BLE001
                import sys, traceback
                traceback.print_exc(file=sys.stderr)
                return 1
        return g

    def cb_begin(_u, qb, kk, share, sample):
        v, x = batch_arrays(qb)
        state["n"] = v.shape[0]
        s = fit(shard.search_begin(v, x, int(kk), int(share)), int(shard.sample_rank(int(kk), int(share))), -np.inf, np.float32)
        out(sample, s.shape, np.float32)[...] = s
        return 0

    def cb_finish(_u, tau, ps, pr, pc):
        n = state["n"]
        s, r, c = shard.search_finish(out(tau, (n,), np.float32).copy())
        out(ps, (n, k), np.float32)[...] = fit(s, k, -np.inf, np.float32)
        out(pr, (n, k), np.int64)[...] = fit(r, k, -1, np.int64)
        out(pc, (n,), np.int32)[...] = c
        return 0

    def cb_search(_u, qb, kk, ps, pr):
        v, x = batch_arrays(qb)
        s, r = shard.search(v, x, int(kk))
        out(ps, (v.shape[0], int(kk)), np.float32)[...] = fit(s, int(kk), -np.inf, np.float32)
        out(pr, (v.shape[0], int(kk)), np.int64)[...] = fit(r, int(kk), -1, np.int64)
        return 0

    def cb_mid_ranks(_u, kk, share, p_local, p_union):
        rl, ru = shard.mid_ranks(int(kk), int(share))
        p_local[0], p_union[0] = int(rl), int(ru)
        return int(rl)

    def cb_mid(_u, tau, r_local, ps):
        n = state["n"]
        s = fit(shard.search_mid(out(tau, (n,), np.float32).copy(), int(r_local)), int(r_local), -np.inf, np.float32)
        out(ps, (n, int(r_local)), np.float32)[...] = s
        return 0

    def cb_pre_ranks(_u, kk, share, p_local, p_union):
        rl, ru = shard.pre_ranks(int(kk), int(share))
        p_local[0], p_union[0] = int(rl), int(ru)
        return int(rl)

    def cb_pre(_u, qb, kk, share, r_local, ps):
        v, x = batch_arrays(qb)
        state["n"], state["share"] = v.shape[0], int(share)
        s = fit(shard.search_pre(v, x, int(kk), int(share), int(r_local)), int(r_local), -np.inf, np.float32)
        out(ps, (v.shape[0], int(r_local)), np.float32)[...] = s
        return 0

    def cb_begin_rest(_u, tau, sample):
        n = state["n"]
        s = fit(shard.search_begin_rest(out(tau, (n,), np.float32).copy()), int(shard.sample_rank(k, state.get("share", 1))), -np.inf, np.float32)
        out(sample, s.shape, np.float32)[...] = s
        return 0

    has_mid = hasattr(shard, "search_mid")          # optional: the second threshold agreement (dhr_search_mid)
    has_pre = hasattr(shard, "search_pre")          # optional: the first agreement in two rounds (dhr_search_pre / dhr_search_begin_rest)
    hs = _lib.HostShard(C.sizeof(_lib.HostShard), 0, None, _lib.HS_SAMPLE_RANK(lambda _u, kk, share: int(shard.sample_rank(int(kk), int(share)))),
                        _lib.HS_UNION_RANK(lambda _u, kk: int(shard.union_rank(int(kk)))), _lib.HS_BEGIN(guard(cb_begin)),
                        _lib.HS_FINISH(guard(cb_finish)), _lib.HS_SEARCH(guard(cb_search)),
                        _lib.HS_MID_RANKS(guard(cb_mid_ranks)) if has_mid else _lib.HS_MID_RANKS(),
                        _lib.HS_MID(guard(cb_mid)) if has_mid else _lib.HS_MID(),
                        _lib.HS_PRE_RANKS(guard(cb_pre_ranks)) if has_pre else _lib.HS_PRE_RANKS(),
                        _lib.HS_PRE(guard(cb_pre)) if has_pre else _lib.HS_PRE(),
                        _lib.HS_BEGIN_REST(guard(cb_begin_rest)) if has_pre else _lib.HS_BEGIN_REST())
    gather = _host_allgather(group) if world > 1 else _lib.ALLGATHER_FN(lambda *_a: 1)
    qb, keep = _lib.make_query_batch(q_value, q_index)
    scores = np.empty((nq, k), np.float32)
    rows = np.empty((nq, k), np.int64)
    _lib.check(lib.dhr_search_sharded_host(C.byref(hs), world, rank, gather, None, C.byref(qb), int(k), scores.ctypes.data, rows.ctypes.data),
               "dhr_search_sharded_host")
    del keep
    return scores, rows


# ----------------------------------------------------------------------------------------------- --PQIP over row shards (config 5)
def _ordered_u32(scores):
    """fp32 tensor -> int64 tensor of order-preserving 32-bit patterns (dhr_internal.h f32_ordered)."""
    import torch
    b = (scores.contiguous() + 0.0).view(torch.int32).to(torch.int64) & 0xFFFFFFFF       # (+ 0.0: -0.0 ties with +0.0, as in the library)
    return torch.where((b >> 31) != 0, (~b) & 0xFFFFFFFF, b | 0x80000000)


def rerank_topk(index, q_value, q_index, rows, k: int):
    """Stage 2 of --PQIP --rerank (gip_retrieval.py:205-215) on the device: exact gated inner product of each query against its candidate
    rows [Q, m] (global rows; < 0 = no candidate), the k best by (score desc, row asc) -- deterministic on exact ties, unlike torch.topk,
    so that the sharded and the unsharded search agree bit for bit.  -> (scores [Q,k] fp32, rows [Q,k] int64; (-inf, -1) padding)."""
    import torch
    s2 = index.score_rows_device(q_value, q_index, rows)
    s2 = torch.where(rows >= 0, s2, torch.full_like(s2, float("-inf")))
    key = ((_ordered_u32(s2) - (1 << 31)) << 32) + (0xFFFFFFFF - rows.clamp(min=0))
    key = torch.where(rows >= 0, key, torch.full_like(key, -(1 << 63)))
    kk = min(k, rows.shape[1])
    best = torch.topk(key, kk, dim=1)
    out_r = torch.gather(rows, 1, best.indices)
    out_s = torch.gather(s2, 1, best.indices)
    if kk < k:
        out_r = torch.cat([out_r, torch.full((rows.shape[0], k - kk), -1, dtype=torch.int64, device=rows.device)], dim=1)
        out_s = torch.cat([out_s, torch.full((rows.shape[0], k - kk), float("-inf"), dtype=torch.float32, device=rows.device)], dim=1)
    return out_s, out_r


def pq_sharded_search(pq, index, q_value, q_index, k1: int, k: int, group=None, n_total: int | None = None):
    """--PQIP --rerank over row shards with the result of the UNSHARDED search (same codebooks; a shard's codes are its slice of the
    corpus' codes): BASELINE config 5, "PQ-quantised index x 8 shards".

    pq / index: this rank's PqIndex / GipIndex (row_offset = the shard's first global row) -- or LISTS of them, one entry per shard, for
    the one-process form (all shards on this process' device; what the single-GPU tests drive).  Steps:
      1. every shard: ADC scan -> its k1 best (score desc, row asc) [dhr_pq_search];
      2. the global cut: theta_q = the k1-th best ADC score of the UNION, found exactly by bisection on the ordered 32-bit pattern of
         the score (32 rounds of: count my scores >= v, sum over the shards) -- no list leaves its shard; ties at theta are taken
         in global row order (shards hold ascending row ranges: an all-gather of the per-shard tie counts gives every shard its quota);
      3. every shard reranks exactly ITS candidates inside the cut [dhr_score_rows] -> its k best;
      4. all-gather of the [Q, k] lists + rank merge [dhr_merge_topk_lists] -> the global [Q, k], identical on every rank.
    Parity of the PQ stage with faiss is unpinned; this function only guarantees sharded == unsharded."""
    import torch
    import torch.distributed as dist
    local = isinstance(pq, (list, tuple))
    pqs, ixs = (list(pq), list(index)) if local else ([pq], [index])
    world = len(pqs) if local else (dist.get_world_size(group) if dist.is_initialized() else 1)

    def allsum(ts):                       # list (one per local shard) of equal-shape tensors -> their sum over ALL shards
        t = torch.stack(ts).sum(0)
        if not local and world > 1:
            dist.all_reduce(t, group=group)
        return t

    def allgather(ts):                    # -> [world, ...] in shard order
        if local:
            return torch.stack(ts)
        if world == 1:
            return ts[0][None]
        out = torch.empty((world,) + tuple(ts[0].shape), dtype=ts[0].dtype, device=ts[0].device)
        dist.all_gather_into_tensor(out.view(world * ts[0].shape[0], *ts[0].shape[1:]) if ts[0].dim() > 0 else out, ts[0].contiguous(), group=group)
        return out

    n_rows = [p.n for p in pqs]
    if n_total is None:
        nt = torch.tensor([sum(n_rows)], dtype=torch.int64, device=torch.device("cuda", pqs[0].device))
        n_total = int(allsum([nt]).item()) if not local else sum(n_rows)
    k1g = min(int(k1), int(n_total))
    # 1. local ADC lists
    lists = []
    for p in pqs:
        s1, r1 = p.search(q_value, min(k1g, p.n), out_device=True)
        lists.append((_ordered_u32(s1), r1))
    nq = lists[0][0].shape[0]
    dev = lists[0][0].device
    # 2. theta = the largest 32-bit pattern v with  #(scores >= v over all shards) >= k1g   (bisection, exact)
    lo = torch.zeros(nq, dtype=torch.int64, device=dev)
    hi = torch.full((nq,), 0xFFFFFFFF, dtype=torch.int64, device=dev)
    for _ in range(33):
        mid = (lo + hi + 1) >> 1
        cnt = allsum([((o >= mid[:, None]) & (r >= 0)).sum(1) for o, r in lists])
        ok = cnt >= k1g
        lo = torch.where(ok, mid, lo)
        hi = torch.where(ok, hi, mid - 1)
    theta = lo
    n_gt = [((o > theta[:, None]) & (r >= 0)).sum(1) for o, r in lists]
    n_eq = [((o == theta[:, None]) & (r >= 0)).sum(1) for o, r in lists]
    need = (k1g - allsum(n_gt)).clamp(min=0)                                   # ties to take, in global row order
    eq_all = allgather(n_eq)                                                    # [world, Q]
    before = torch.cumsum(eq_all, 0) - eq_all                                   # ties held by the shards in front
    rank0 = 0 if local else (dist.get_rank(group) if dist.is_initialized() else 0)
    # 3. exact rerank of the shard's candidates inside the cut
    outs = []
    for i, ((o, r), ix) in enumerate(zip(lists, ixs)):
        me = i if local else rank0
        quota = (need - before[me]).clamp(min=0)
        take = n_gt[i] + torch.minimum(quota, n_eq[i])                          # a prefix of the sorted list (ties are in row order)
        pos = torch.arange(r.shape[1], device=dev)[None, :]
        cand = torch.where(pos < take[:, None], r, torch.full_like(r, -1))
        outs.append(rerank_topk(ix, q_value, q_index, cand, k))
    # 4. reduce
    if local:
        return merge_sorted_lists(torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), k)
    return allgather_merge(outs[0][0], outs[0][1], k, group)
