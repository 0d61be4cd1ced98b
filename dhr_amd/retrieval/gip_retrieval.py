"""MI355X drop-in for castorini/dhr's retrieval/gip_retrieval.py (brute-force dense-hybrid search).

Same CLI flags (incl. the typos --shrad/--total_shrad/--lamda), same pickle inputs
([value fp16, index u8|i8|i16|None|0, ids]), same `result.trec` output and the same function
signatures as the reference (/root/reference/retrieval/gip_retrieval.py:60-165, 233-344) -- but the
per-query torch loop (mask * corpus -> einsum -> topk, :115-126) is replaced by ONE call into
libdhr_hip.so per query batch: a bound GEMM on the matrix cores with a fused threshold filter, exact
fp64 rescoring of the survivors and a per-query top-k merge, all on the GPU (dhr_amd/csrc).

There is no CPU path here.  If the HIP library or a GPU is missing every entry point raises.
"""
from __future__ import annotations

import argparse
import os
import ctypes as C
import pickle
import time

import numpy as np

from .. import _lib


# ----------------------------------------------------------------------------------------------- index handle
class GipIndex:
    """One corpus shard resident on one GPU (dhr_index_create).  value: fp16 [N, K] numpy array or
    torch tensor (host or device); index: [N, D_dlr] uint8/int8/int16 or None (dense-only)."""

    def __init__(self, value, index=None, *, emb_dim=None, device: int = 0, row_offset: int = 0, idx_buckets: int = 0):
        lib = _lib.load()
        value = _as_f16(value)
        n, k = int(value.shape[0]), int(value.shape[1])
        if index is not None and np.isscalar(index):          # merged dense index stores the int 0 (index.py:40-43)
            index = None
        d_dlr = 0 if index is None else int(index.shape[1])
        if index is not None and emb_dim is not None and emb_dim != d_dlr:
            raise ValueError(f"--emb_dim {emb_dim} does not match the index array width {d_dlr}")
        # (--emb_dim that is not a multiple of 8: the library appends zero slices to its own copies of the corpus and of every
        # query batch -- dhr_index_create; until round 3 this mirror padded the arrays on the host)
        desc = _lib.IndexDesc()
        desc.device = device
        desc.n_rows = n
        desc.d_dlr, desc.d_cls = d_dlr, k - d_dlr
        desc.value, desc.ld_value, desc.mem_kind = _lib._ptr_ld(value)
        if index is not None:
            p, ld, kind = _lib._ptr_ld(index)
            if kind != desc.mem_kind:
                raise ValueError("corpus value and index must live in the same memory kind")
            desc.index, desc.ld_index, desc.index_dtype = p, ld, _lib.idx_code(index.dtype)
        else:
            desc.index, desc.ld_index, desc.index_dtype = None, 0, _lib.IDX_NONE
        desc.row_offset = row_offset
        desc.idx_buckets = idx_buckets
        h = C.c_void_p()
        _lib.check(lib.dhr_index_create(C.byref(desc), C.byref(h)), "dhr_index_create")
        self._h, self._lib = h, lib
        self._pending = None
        self.n_rows, self.k, self.d_dlr, self.device, self.row_offset = n, k, d_dlr, device, row_offset

    def _qb(self, q_value, q_index):
        """Query batch for the library (the caller's own record width; the library pads where --emb_dim is not a multiple of 8).
        The C struct carries leading dimensions, not widths: a batch of another width than the index is refused here (the reference's einsum
        raises on it, gip_retrieval.py:121), instead of being read as the first K columns of wider rows."""
        if len(q_value.shape) != 2 or int(q_value.shape[1]) != self.k:
            raise ValueError(f"query batch of shape {tuple(q_value.shape)} against an index of width {self.k}")
        if q_index is not None and (len(q_index.shape) != 2 or int(q_index.shape[1]) != self.d_dlr or int(q_index.shape[0]) != int(q_value.shape[0])):
            raise ValueError(f"query index array of shape {tuple(q_index.shape)} against {int(q_value.shape[0])} queries and an index array of width {self.d_dlr}")
        return _lib.make_query_batch(q_value, q_index)

    # ---- device-ready index file (dhr_index_save / dhr_index_load)
    def save(self, path: str, docids=None):
        """Write the built device images to `path`; `docids` (the third element of the reference's index record)
        travels as the file's blob so that one file replaces the pickle."""
        blob = pickle.dumps(list(docids), protocol=4) if docids is not None else b""
        _lib.check(self._lib.dhr_index_save(self._h, os.fsencode(path), blob if blob else None, len(blob)), "dhr_index_save")

    @staticmethod
    def is_device_file(path: str) -> bool:
        try:
            with open(path, "rb") as f:
                return f.read(8) == _lib.FILE_MAGIC
        except OSError:
            return False

    @classmethod
    def load(cls, path: str, device: int = 0, row_offset: int = -1):
        """-> (GipIndex, docids or None): mmap + stream the sections to the device, no re-layout."""
        lib = _lib.load()
        info = _lib.FileInfo()
        _lib.check(lib.dhr_index_file_info(os.fsencode(path), C.byref(info)), "dhr_index_file_info")
        h = C.c_void_p()
        _lib.check(lib.dhr_index_load(os.fsencode(path), device, row_offset, C.byref(h)), "dhr_index_load")
        self = cls.__new__(cls)
        self._h, self._lib = h, lib
        self._pending = None
        self.n_rows, self.k, self.d_dlr, self.device = int(info.n_rows), int(info.d_dlr + info.d_cls), int(info.d_dlr), device
        self.row_offset = int(info.row_offset) if row_offset < 0 else row_offset
        docids = None
        if info.blob_bytes > 0:
            with open(path, "rb") as f:
                f.seek(info.blob_offset)
                docids = pickle.loads(f.read(info.blob_bytes))
        return self, docids

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dhr_index_destroy(self._h)
            self._h = None

    __del__ = close

    def set_param(self, param: int, value: int):
        _lib.check(self._lib.dhr_index_set_param(self._h, param, int(value)), "dhr_index_set_param")

    def device_bytes(self) -> int:
        return int(self._lib.dhr_index_device_bytes(self._h))

    def info(self, what: int) -> float:
        """dhr_index_get_info: read-only facts about the built index (_lib.INFO_*)."""
        out = C.c_double()
        _lib.check(self._lib.dhr_index_get_info(self._h, what, C.byref(out)), "dhr_index_get_info")
        return out.value

    def stats(self) -> dict:
        st = _lib.SearchStats()
        _lib.check(self._lib.dhr_get_stats(self._h, C.byref(st)), "dhr_get_stats")
        return st.as_dict()

    def search(self, q_value, q_index, k: int, *, out_device: bool = False, stream: int = 0):
        """-> (scores fp32 [Q,k], rows int64 [Q,k]); rows are global (row_offset added), best first,
        (-inf, -1) padding when k > n_rows.  numpy outputs unless out_device (then torch cuda tensors)."""
        qb, keep = self._qb(q_value, q_index)
        nq = qb.n_queries
        if out_device:
            import torch
            dev = torch.device("cuda", self.device)
            scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
            rows = torch.empty((nq, k), dtype=torch.int64, device=dev)
            ps, pr, kind = scores.data_ptr(), rows.data_ptr(), _lib.MEM_DEVICE
        else:
            scores = np.empty((nq, k), np.float32)
            rows = np.empty((nq, k), np.int64)
            ps, pr, kind = scores.ctypes.data, rows.ctypes.data, _lib.MEM_HOST
        _lib.check(self._lib.dhr_search(self._h, C.byref(qb), int(k), ps, pr, kind, stream), "dhr_search")
        del keep
        return scores, rows

    def search_rerank(self, q1_value, q1_index, q_value, q_index, k1: int, k: int, *, stream: int = 0):
        """Two-stage search on the device (dhr_search_rerank): top-k1 of the stage-1 batch, exact GIP of the full
        batch on those rows, top-k of that.  -> (scores fp32 [Q,k], rows int64 [Q,k]) numpy."""
        qb1, keep1 = self._qb(q1_value, q1_index)
        qb2, keep2 = self._qb(q_value, q_index)
        nq = qb1.n_queries
        scores = np.empty((nq, k), np.float32)
        rows = np.empty((nq, k), np.int64)
        _lib.check(self._lib.dhr_search_rerank(self._h, C.byref(qb1), C.byref(qb2), int(k1), int(k), scores.ctypes.data,
                                               rows.ctypes.data, _lib.MEM_HOST, stream), "dhr_search_rerank")
        del keep1, keep2
        return scores, rows

    # ---- staged search (row-sharded path, dhr_amd/dist.py)
    def sample_rank(self, k: int) -> int:
        return int(self._lib.dhr_search_sample_rank(self._h, int(k)))

    def union_rank(self, k: int) -> int:
        """Rank of the union of the shards' samples that defines the common threshold (dhr_search_union_rank)."""
        return int(self._lib.dhr_search_union_rank(self._h, int(k)))

    def search_begin(self, q_value, q_index, k: int, stream: int = 0):
        """Runs the sampled part; -> torch cuda tensor [Q, r] with this shard's r best sample scores
        (None when the shard is too small to sample: then the whole search already ran)."""
        import torch
        qb, keep = self._qb(q_value, q_index)
        r = self.sample_rank(k)
        out = None
        if r > 0:
            out = torch.empty((qb.n_queries, r), dtype=torch.float32, device=torch.device("cuda", self.device))
        _lib.check(self._lib.dhr_search_begin(self._h, C.byref(qb), int(k), out.data_ptr() if out is not None else None,
                                              stream), "dhr_search_begin")
        self._pending = (qb.n_queries, int(k))
        del keep
        return out

    def _staged(self, what: str):
        """(n_queries, k) of the staged search this handle has open (search_begin / search_pre); the later stages size their outputs by it."""
        if getattr(self, "_pending", None) is None:
            raise _lib.DhrError(f"{what} without a matching search_begin / search_pre on this handle", _lib.ERR_INVALID)
        return self._pending

    def pre_ranks(self, k: int):
        """(local, union) ranks of the first agreement's first round (dhr_search_pre_ranks); (0, 0): this index has no pre step."""
        lo, un = C.c_int32(), C.c_int32()
        self._lib.dhr_search_pre_ranks(self._h, int(k), C.byref(lo), C.byref(un))
        return int(lo.value), int(un.value)

    def search_pre(self, q_value, q_index, k: int, r_local: int = 0, stream: int = 0):
        """Query preparation + the first part of the sampled run; -> torch cuda tensor [Q, r_local]: this shard's best sample scores seen so
        far (dhr_search_pre).  search_begin_rest then takes the threshold the shards agree on from them."""
        import torch
        qb, keep = self._qb(q_value, q_index)
        rl = int(r_local) or self.pre_ranks(k)[0]
        out = torch.empty((qb.n_queries, rl), dtype=torch.float32, device=torch.device("cuda", self.device))
        _lib.check(self._lib.dhr_search_pre(self._h, C.byref(qb), int(k), rl, out.data_ptr(), stream), "dhr_search_pre")
        self._pending = (qb.n_queries, int(k))
        del keep
        return out

    def search_begin_rest(self, tau, stream: int = 0):
        """The rest of the sampled run, filtered at the agreed thresholds tau [Q]; -> what search_begin returns (dhr_search_begin_rest)."""
        import torch
        nq, k = self._staged("search_begin_rest")
        dev = torch.device("cuda", self.device)
        out = torch.empty((nq, self.sample_rank(k)), dtype=torch.float32, device=dev)
        tau = tau.to(device=dev, dtype=torch.float32).contiguous()
        _lib.check(self._lib.dhr_search_begin_rest(self._h, tau.data_ptr(), out.data_ptr(), stream), "dhr_search_begin_rest")
        return out

    def mid_ranks(self, k: int):
        """(local, union) ranks of the second threshold agreement (dhr_search_mid_ranks); (0, 0): this index has no mid step."""
        lo, un = C.c_int32(), C.c_int32()
        self._lib.dhr_search_mid_ranks(self._h, int(k), C.byref(lo), C.byref(un))
        return int(lo.value), int(un.value)

    def search_mid(self, tau_hat, r_local: int = 0, stream: int = 0):
        """First slice of the main pass with the common thresholds tau_hat [Q]; -> torch cuda tensor [Q, r_local]: this shard's best
        scores seen so far (dhr_search_mid).  search_finish then takes the thresholds of the second agreement."""
        import torch
        nq, k = self._staged("search_mid")
        dev = torch.device("cuda", self.device)
        rl = int(r_local) or self.mid_ranks(k)[0]
        out = torch.empty((nq, rl), dtype=torch.float32, device=dev)
        tau_hat = tau_hat.to(device=dev, dtype=torch.float32).contiguous()
        _lib.check(self._lib.dhr_search_mid(self._h, tau_hat.data_ptr(), rl, out.data_ptr(), stream), "dhr_search_mid")
        return out

    def search_finish(self, tau_hat, stream: int = 0):
        """tau_hat: torch cuda tensor [Q] (or None after a non-sampled begin).  -> (scores [Q,k], rows [Q,k],
        count [Q] int32 rows reaching tau_hat, -1 = list overflow), all torch cuda tensors."""
        import torch
        nq, k = self._staged("search_finish")
        dev = torch.device("cuda", self.device)
        scores = torch.empty((nq, k), dtype=torch.float32, device=dev)
        rows = torch.empty((nq, k), dtype=torch.int64, device=dev)
        count = torch.empty((nq,), dtype=torch.int32, device=dev)
        tp = None
        if tau_hat is not None:
            tau_hat = tau_hat.to(device=dev, dtype=torch.float32).contiguous()
            tp = tau_hat.data_ptr()
        _lib.check(self._lib.dhr_search_finish(self._h, tp, scores.data_ptr(), rows.data_ptr(), count.data_ptr(),
                                               _lib.MEM_DEVICE, stream), "dhr_search_finish")
        return scores, rows, count

    def score_rows(self, q_value, q_index, rows):
        """Exact gated inner product of each query against its own list of (global) rows [Q, m]."""
        qb, keep = self._qb(q_value, q_index)
        rows = np.ascontiguousarray(rows, np.int64)
        out = np.empty(rows.shape, np.float32)
        _lib.check(self._lib.dhr_score_rows(self._h, C.byref(qb), int(rows.shape[1]), rows.ctypes.data, out.ctypes.data,
                                            _lib.MEM_HOST, 0), "dhr_score_rows")
        del keep
        return out

    def score_rows_device(self, q_value, q_index, rows):
        """score_rows for a torch cuda int64 tensor of rows [Q, m]: -> torch cuda fp32 [Q, m] (no host round trip)."""
        import torch
        qb, keep = self._qb(q_value, q_index)
        rows = rows.contiguous()
        out = torch.empty(rows.shape, dtype=torch.float32, device=rows.device)
        _lib.check(self._lib.dhr_score_rows(self._h, C.byref(qb), int(rows.shape[1]), rows.data_ptr(), out.data_ptr(), _lib.MEM_DEVICE, 0), "dhr_score_rows")
        del keep
        return out


def _as_f16(a):
    """Corpus values are fp16 on disk; the reference widens them to fp32 on its CPU path
    (gip_retrieval.py:313).  Accept either and hand fp16 to the library (the widening is exact, so
    narrowing an fp32 array that came from the file is too; anything else is rejected)."""
    name = str(a.dtype).replace("torch.", "")
    if name == "float16":
        return a
    if name != "float32":
        raise TypeError(f"corpus values must be float16 (or float32 widened from float16), got {a.dtype}")
    if isinstance(a, np.ndarray):
        h = a.astype(np.float16)
        if not np.array_equal(h.astype(np.float32), a):
            raise ValueError("fp32 corpus values are not fp16-representable")
        return h
    h = a.half()
    if not bool((h.float() == a).all()):
        raise ValueError("fp32 corpus values are not fp16-representable")
    return h


def _np(a):
    return a if isinstance(a, np.ndarray) else a.detach().cpu().numpy()


def _to_dicts(qids, rows, scores, offset=0, args=None):
    keep_arrays = getattr(args, "_result_arrays", None)
    if keep_arrays is not None:        # main(): the run file is written from the arrays by the library (write_trec_native), no dicts
        keep_arrays.append((np.ascontiguousarray(rows, np.int64), np.ascontiguousarray(scores, np.float32), int(offset)))
        return {}, {}
    all_results, all_scores = {}, {}
    for i, qid in enumerate(qids):
        r = rows[i]
        keep = r >= 0
        all_results[qid] = (r[keep] - offset).tolist()
        all_scores[qid] = scores[i][keep].tolist()
    return all_results, all_scores


QUERY_CHUNK = 8192      # queries per library call: the search workspace grows linearly with the batch (the reference loops per query and takes any number)


def _by_chunks(n_queries, call):
    """call(lo, hi) -> (scores, rows) numpy arrays of queries [lo, hi); the whole query set in slices of QUERY_CHUNK."""
    if n_queries <= QUERY_CHUNK:
        return call(0, n_queries)
    parts = [call(lo, min(n_queries, lo + QUERY_CHUNK)) for lo in range(0, n_queries, QUERY_CHUNK)]
    return np.concatenate([p[0] for p in parts], axis=0), np.concatenate([p[1] for p in parts], axis=0)


def _sl(a, lo, hi):
    return None if a is None else a[lo:hi]


def _corpus_index(corpus_embs, corpus_arg_idxs, args):
    if isinstance(corpus_embs, GipIndex):
        return corpus_embs, False
    return GipIndex(corpus_embs, corpus_arg_idxs, device=getattr(args, "device", 0)), True


# ----------------------------------------------------------------------------------------------- reference API
def IP_retrieval(qids, query_embs, corpus_embs, args):
    """gip_retrieval.py:60-85: plain inner product, k best rows (local indices), best first.
    corpus_embs: array/tensor [N,K] or a prebuilt GipIndex (dense-only)."""
    index, owned = _corpus_index(corpus_embs, None, args)
    start_time = time.time()
    scores, rows = _by_chunks(len(qids), lambda lo, hi: index.search(query_embs[lo:hi], None, args.topk))
    res = _to_dicts(qids, rows, scores, index.row_offset, args)
    time_per_query = (time.time() - start_time) / len(qids)
    print('Retrieving {} queries ({:0.3f} s/query), average number of index use {}'.format(len(qids), time_per_query, 0.0))
    if owned:
        index.close()
    return res


def GIP_retrieval(qids, query_embs, query_arg_idxs, corpus_embs, corpus_arg_idxs, args):
    """gip_retrieval.py:88-165.  brute force / theta==0: exact gated inner product over all rows.
    theta>0: stage 1 keeps only the query columns with value > theta (:130-136) or uses the plain
    inner product (--IP, :139); --rerank rescans the agip_topk stage-1 rows with the exact GIP
    (:141-153).  Row indices are local to the corpus slice, as in the reference."""
    index, owned = _corpus_index(corpus_embs, corpus_arg_idxs, args)
    theta = 0 if args.brute_force else args.theta
    n = index.n_rows
    start_time = time.time()
    total_num_idx = 0
    try:
        if theta == 0:
            if args.topk > n and not getattr(args, "allow_short", False):
                raise RuntimeError("selected index k out of range")          # torch.topk, :123
            total_num_idx = args.emb_dim * len(qids)
            scores, rows = _by_chunks(len(qids), lambda lo, hi: index.search(query_embs[lo:hi], _sl(query_arg_idxs, lo, hi), args.topk))
        else:
            q = _np(query_embs).astype(np.float32)
            qi = _np(query_arg_idxs)
            k1 = args.agip_topk if args.rerank else args.topk
            if k1 > n and not getattr(args, "allow_short", False):
                raise RuntimeError("selected index k out of range")
            if not args.IP:
                q1, qi1 = np.where(q > theta, q, np.float32(0)), qi           # restrict to the important columns
            else:
                q1, qi1 = q, None                                             # ungated inner product
            def two_stage(lo, hi):
                if args.rerank:
                    return index.search_rerank(q1[lo:hi], _sl(qi1, lo, hi), q[lo:hi], qi[lo:hi], k1, min(args.topk, k1))   # both stages on the device
                return index.search(q1[lo:hi], _sl(qi1, lo, hi), k1)
            scores, rows = _by_chunks(len(qids), two_stage)
        res = _to_dicts(qids, rows, scores, index.row_offset, args)
    finally:
        if owned:
            index.close()
    average_num_idx = total_num_idx / len(qids)
    time_per_query = (time.time() - start_time) / len(qids)
    print('Retrieving {} queries ({:0.3f} s/query), average number of index use {}'.format(len(qids), time_per_query, average_num_idx))
    return res


def PQ_IP_retrieval(qids, query_embs, query_arg_idxs, corpus_embs, corpus_arg_idxs, args):
    """gip_retrieval.py:167-231: first stage = product-quantised inner product over the WHOLE vector (no gate), agip_topk
    candidates (faiss IndexPQ.search, :202); --rerank: exact GIP of those candidates, top-k (:205-215); otherwise the first topk
    PQ results (:218-221).  `--faiss_pq_index_path` is a faiss IndexPQ file (retrieval/quantize_index.py, the reference's or this
    build's).  The codes stay resident at one byte per sub-quantiser and row and are searched by the ADC scan (dhr_pq_search);
    the candidates are rescored exactly by dhr_score_rows.  Parity with faiss unpinned (faiss is not part of the reference tree)."""
    from . import quantize_index as QI
    assert args.faiss_pq_index_path is not None, 'you do not spesify your PQ index through --faiss_pq_index_path'
    print('Load PQ index ...')
    pq = QI.load_pq(args.faiss_pq_index_path)
    index, owned = _corpus_index(corpus_embs, corpus_arg_idxs, args)
    if pq["codes"].shape[0] != index.n_rows or pq["d"] != index.k:
        raise ValueError("the PQ index does not describe the same corpus as --index_path")
    pq_index = QI.PqIndex(pq["codebooks"], pq["codes"], nbits=pq["nbits"], device=getattr(args, "device", 0), row_offset=index.row_offset)
    start_time = time.time()
    try:
        q = _np(query_embs).astype(np.float32)
        k1 = min(args.agip_topk, index.n_rows)
        qi_np = _np(query_arg_idxs)

        def pq_stage(lo, hi):
            s1, r1 = pq_index.search(q[lo:hi], k1)
            if not args.rerank:
                return s1[:, : args.topk], r1[:, : args.topk]
            s2 = index.score_rows(q[lo:hi], qi_np[lo:hi], r1)
            order = np.lexsort((r1, -s2.astype(np.float64)), axis=1)[:, : args.topk]
            return np.take_along_axis(s2, order, axis=1), np.take_along_axis(r1, order, axis=1)
        scores, rows = _by_chunks(len(qids), pq_stage)
        res = _to_dicts(qids, rows, scores, index.row_offset, args)
    finally:
        pq_index.close()
        if owned:
            index.close()
    time_per_query = (time.time() - start_time) / len(qids)
    print('Retrieving {} queries ({:0.3f} s/query)'.format(len(qids), time_per_query))
    return res


# ----------------------------------------------------------------------------------------------- main()
def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("--query_emb_path", type=str, required=True)
    parser.add_argument("--index_path", type=str, required=True)
    parser.add_argument("--faiss_pq_index_path", type=str, default=None)
    parser.add_argument("--emb_dim", type=int, default=768, help='DLR dimension')
    parser.add_argument("--theta", type=float, default=0.1)
    parser.add_argument("--topk", type=int, default=1000)
    parser.add_argument("--agip_topk", type=int, default=10000)
    parser.add_argument("--combine_cls", action='store_true')
    parser.add_argument("--IP", action='store_true')
    parser.add_argument("--PQIP", action='store_true')
    parser.add_argument("--batch", type=int, default=1)
    parser.add_argument("--brute_force", action='store_true')
    parser.add_argument("--use_gpu", action='store_true')
    parser.add_argument("--rerank", action='store_true')
    parser.add_argument("--lamda", type=float, default=1, help='weight for [CSL] for concatenation')
    parser.add_argument("--total_shrad", type=int, default=1)
    parser.add_argument("--shrad", type=int, default=0)
    parser.add_argument("--run_name", type=str, default='h2oloo')
    # not in the reference: which GPU holds the shard (the reference hard-wires cuda:0, :261)
    parser.add_argument("--device", type=int, default=0)
    parser.add_argument("--save_device_index", type=str, default=None,
                        help="(not in the reference) also write the built index as a device-ready file; pass that file as "
                             "--index_path later to skip the pickle load and the re-layout")
    parser.add_argument("--output", type=str, default=None, help="override the result file name")
    return parser


def load_queries(path, emb_dim, lamda):
    """gip_retrieval.py:263-283: [value fp16, index|None, qids]; fp32; CLS tail *= lamda in fp32."""
    with open(path, 'rb') as f:
        query_embs, query_arg_idxs, qids = pickle.load(f)
    query_embs = np.asarray(query_embs).astype(np.float32)
    if query_arg_idxs is None or np.isscalar(query_arg_idxs):
        query_arg_idxs = None
    else:
        query_arg_idxs = np.asarray(query_arg_idxs)
    cls_dim = query_embs.shape[1] - emb_dim
    if cls_dim > 0:
        query_embs[:, -cls_dim:] = np.float32(lamda) * query_embs[:, -cls_dim:]
    return query_embs, query_arg_idxs, qids


def shard_bounds(n_docs, total_shrad, shrad):
    """gip_retrieval.py:292-306: per = len(docids)//total_shrad; the last shard takes the remainder."""
    per = n_docs // total_shrad
    lo = per * shrad
    hi = n_docs if shrad == total_shrad - 1 else per * (shrad + 1)
    return lo, hi


def load_corpus_shard(path, total_shrad=1, shrad=0):
    """gip_retrieval.py:287-306.  No fp32 copy is made; the fp16 slice goes straight to the GPU."""
    with open(path, 'rb') as f:
        corpus_embs, corpus_arg_idxs, docids = pickle.load(f)
    lo, hi = shard_bounds(len(docids), total_shrad, shrad)
    corpus_embs = np.asarray(corpus_embs)[lo:hi]
    if corpus_arg_idxs is None or np.isscalar(corpus_arg_idxs):
        corpus_arg_idxs = None                                                # dense index (None or merged 0)
    else:
        corpus_arg_idxs = np.asarray(corpus_arg_idxs)[lo:hi]
    return corpus_embs, corpus_arg_idxs, docids[lo:hi], lo


def write_trec(fout, results, scores, docids, run_name):
    """gip_retrieval.py:333-342 (self-matches skipped, rank numbers keep their gaps)."""
    for query_id in results:
        result = results[query_id]
        score = scores[query_id]
        for rank, docidx in enumerate(result):
            docid = docids[docidx]
            if docid != query_id:
                fout.write('{} Q0 {} {} {} {}\n'.format(query_id, docid, rank + 1, score[rank], run_name))


def _id_blob(ids):
    """list of str -> (bytes: the ids with one newline after each, int64 offsets [n+1]) for dhr_write_trec, or None when the ids are
    not plain newline-free strings (the Python writer then reproduces the reference's formatting of whatever they are)."""
    if not all(type(x) is str for x in ids):
        return None
    blob = ("\n".join(ids) + "\n").encode("utf-8") if len(ids) else b""
    ends = np.flatnonzero(np.frombuffer(blob, np.uint8) == 10)
    if len(ends) != len(ids):
        return None                                   # an id contains a newline
    off = np.zeros(len(ids) + 1, np.int64)
    off[1:] = ends + 1
    return blob, off


def write_trec_native(path, qids, rows, scores, row_base, docids, run_name, append=False):
    """The writer loop of gip_retrieval.py:333-342 in the library (dhr_write_trec: formatted on the host's cores, byte-identical to
    write_trec below).  rows: int64 [Q, k] (negative = padding), scores float32 [Q, k].  Returns the number of lines, or None when
    the ids cannot be handed over as byte blobs (then nothing was written)."""
    qb, db = _id_blob(list(qids)), _id_blob(docids)
    if qb is None or db is None:
        return None
    lib = _lib.load()
    rows = np.ascontiguousarray(rows, np.int64)
    scores = np.ascontiguousarray(scores, np.float32)
    lines = C.c_int64()
    _lib.check(lib.dhr_write_trec(os.fsencode(path), 1 if append else 0, rows.shape[0], rows.shape[1], qb[0], qb[1].ctypes.data, db[0], db[1].ctypes.data,
                                  len(docids), rows.ctypes.data, int(row_base), scores.ctypes.data, run_name.encode("utf-8"), 1, 0, C.byref(lines)),
               "dhr_write_trec")
    return int(lines.value)


def main(argv=None):
    args = build_parser().parse_args(argv)
    args._result_arrays = []
    _lib.load()                                      # fail before touching the data if the HIP library is missing
    print('Load query embeddings ...')
    query_embs, query_arg_idxs, qids = load_queries(args.query_emb_path, args.emb_dim, args.lamda)
    print('Load index ...')
    if GipIndex.is_device_file(args.index_path):
        # device-ready file written by --save_device_index: no pickle, no re-layout (one file = one shard)
        if args.total_shrad != 1:
            raise ValueError("a device-ready index file holds exactly one shard: write one file per shard "
                             "(--save_device_index together with --shrad/--total_shrad) and run without --total_shrad")
        index, docids = GipIndex.load(args.index_path, device=args.device)
        if docids is None:
            raise ValueError("the device-ready index file carries no docid list")
        if (query_arg_idxs is not None) != (index.d_dlr > 0):
            raise ValueError("query file and index disagree about the slice-index array")
        if query_arg_idxs is not None and args.emb_dim != index.d_dlr:
            raise ValueError(f"--emb_dim {args.emb_dim} does not match the index array width {index.d_dlr}")
    else:
        corpus_embs, corpus_arg_idxs, docids, _lo = load_corpus_shard(args.index_path, args.total_shrad, args.shrad)
        if query_arg_idxs is not None and corpus_arg_idxs is None:
            raise ValueError("the query file has an index array but the corpus index has none")
        if query_arg_idxs is not None:
            index = GipIndex(corpus_embs, corpus_arg_idxs, emb_dim=args.emb_dim, device=args.device)
        else:
            index = GipIndex(corpus_embs, None, device=args.device)
        if args.save_device_index:
            index.save(args.save_device_index, docids)
    if query_arg_idxs is not None:
        if not args.PQIP:
            results, scores = GIP_retrieval(qids, query_embs, query_arg_idxs, index, None, args)
        else:
            results, scores = PQ_IP_retrieval(qids, query_embs, query_arg_idxs, index, None, args)
    else:
        results, scores = IP_retrieval(qids, query_embs, index, args)
    index.close()
    if args.output:
        name = args.output
    elif args.total_shrad == 1:
        name = 'result.trec'
    else:
        name = 'result{}.trec'.format(args.shrad)
    rows_a, scores_a, base = args._result_arrays[-1]
    if write_trec_native(name, qids, rows_a, scores_a, base, docids, args.run_name) is None:
        with open(name, 'w') as fout:                 # ids that are not plain strings: the reference's own loop
            write_trec(fout, *_to_dicts(qids, rows_a, scores_a, base), docids, args.run_name)
    print('finish')


if __name__ == "__main__":
    main()
