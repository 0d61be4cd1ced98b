"""ctypes binding of libdhr_hip.so (include/dhr_hip.h).  There is no CPU fallback: if the library
is missing or a call fails this raises -- the product path is the HIP path or nothing."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _build

DHR_OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_INTERNAL, ERR_NOMEM, ERR_PEER = -1, -2, -3, -4, -5, -6
ABI_INDEX_DESC, ABI_QUERY_BATCH, ABI_SEARCH_STATS, ABI_FILE_INFO, ABI_HOST_SHARD = 0, 1, 2, 3, 4
IDX_NONE, IDX_U8, IDX_I8, IDX_I16 = 0, 1, 2, 3
VAL_F16, VAL_F32 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
PARAM_CAND_CAP, PARAM_FIRST_ROWS, PARAM_PROFILE, PARAM_MAX_GROWTH, PARAM_SAMPLE_PERIOD, PARAM_GEMM_VARIANT, PARAM_MAIN_CHUNKS, PARAM_PROGRESSIVE_THR, PARAM_AUX_CUS, PARAM_GEMM_EXCLUSIVE, PARAM_OVERLAP_AUX, PARAM_SAMPLE_SHARE, PARAM_ASYNC_CONTROLLER, PARAM_LIST_STRIDE = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14
OPT_DENSE_I8, OPT_GATED_I8 = 1, 2
COMM_TRANSPORT, COMM_WORLD, COMM_RANK, COMM_DEVICE = 0, 1, 2, 3
INFO_DENSE_I8, INFO_I8_SCALE, INFO_I8_ROW_ERR, INFO_I8_ROW_NORM, INFO_ROW_NORM_MAX, INFO_TILE_BYTES, INFO_GATED_I8, INFO_GEMM_KERNEL = 1, 2, 3, 4, 5, 6, 7, 8

EXPORTS = ["dhr_version", "dhr_abi_sizes", "dhr_abi_size", "dhr_debug_fail_alloc", "dhr_set_option", "dhr_index_get_info", "dhr_last_error", "dhr_index_create", "dhr_index_destroy", "dhr_index_set_param",
           "dhr_index_device_bytes", "dhr_search", "dhr_score_rows", "dhr_merge_topk", "dhr_merge_topk_host", "dhr_merge_topk_lists", "dhr_merge_topk_lists_host",
           "dhr_get_stats", "dhr_debug_bound_scores", "dhr_debug_query_margins", "dhr_debug_gemm_time", "dhr_debug_seq_to_tile", "dhr_debug_sharded_repairs", "dhr_search_sample_rank", "dhr_search_union_rank", "dhr_search_begin",
           "dhr_search_finish", "dhr_search_mid_ranks", "dhr_search_mid", "dhr_search_pre_ranks", "dhr_search_pre", "dhr_search_begin_rest", "dhr_search_rerank", "dhr_comm_unique_id", "dhr_comm_create", "dhr_comm_wrap", "dhr_comm_create_callback", "dhr_comm_destroy", "dhr_comm_info", "dhr_comm_abort", "dhr_search_sharded", "dhr_search_sharded_local", "dhr_search_sharded_host", "dhr_pq_create", "dhr_pq_destroy", "dhr_pq_device_bytes", "dhr_pq_search", "dhr_pq_adc_scores", "dhr_pq_last_scan", "dhr_index_save", "dhr_index_file_info", "dhr_index_load", "dhr_densify", "dhr_pq_train", "dhr_pq_encode", "dhr_pq_decode", "dhr_pq_train_nbits", "dhr_pq_encode_nbits", "dhr_pq_decode_nbits", "dhr_write_trec", "dhr_format_float"]


class DhrError(RuntimeError):
    """A failed C-ABI call.  `status` is the negative dhr_status (None when the failure is the binding's own)."""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


class IndexDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("mem_kind", C.c_int32), ("n_rows", C.c_int64), ("d_dlr", C.c_int32),
                ("d_cls", C.c_int32), ("value", C.c_void_p), ("ld_value", C.c_int64), ("index", C.c_void_p),
                ("index_dtype", C.c_int32), ("idx_buckets", C.c_int32), ("ld_index", C.c_int64),
                ("row_offset", C.c_int64)]


class QueryBatch(C.Structure):
    _fields_ = [("n_queries", C.c_int32), ("mem_kind", C.c_int32), ("value", C.c_void_p), ("value_dtype", C.c_int32),
                ("index_dtype", C.c_int32), ("ld_value", C.c_int64), ("index", C.c_void_p), ("ld_index", C.c_int64)]


class SearchStats(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_queries", C.c_int64), ("k", C.c_int64), ("phases", C.c_int32),
                ("overflow_retries", C.c_int32), ("candidates_bound", C.c_int64), ("candidates_exact", C.c_int64),
                ("gemm_rows", C.c_int64), ("sample_fallback_queries", C.c_int64), ("gemm_ms", C.c_double), ("refine_ms", C.c_double),
                ("rescore_ms", C.c_double), ("select_ms", C.c_double), ("prep_ms", C.c_double),
                ("total_ms", C.c_double), ("gemm_flops", C.c_double), ("gemm_flops_alg", C.c_double)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


class FileInfo(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("row_offset", C.c_int64), ("d_dlr", C.c_int32), ("d_cls", C.c_int32),
                ("index_dtype", C.c_int32), ("idx_buckets", C.c_int32), ("file_version", C.c_uint32), ("reserved", C.c_uint32),
                ("payload_bytes", C.c_int64), ("blob_offset", C.c_int64), ("blob_bytes", C.c_int64)]


FILE_MAGIC = b"DHRIDX1\x00"

# dhr_allgather_fn / dhr_host_shard (include/dhr_hip.h): the caller's transport and, for the CPU tests, the caller's shard
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
HS_SAMPLE_RANK = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32)
HS_UNION_RANK = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32)
HS_BEGIN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_int32, C.c_void_p)
HS_FINISH = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
HS_SEARCH = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_void_p, C.c_void_p)
HS_MID_RANKS = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32))
HS_MID = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p)
HS_PRE_RANKS = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32))
HS_PRE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_int32, C.c_int32, C.c_void_p)
HS_BEGIN_REST = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)


class HostShard(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved", C.c_uint32), ("user", C.c_void_p), ("sample_rank", HS_SAMPLE_RANK), ("union_rank", HS_UNION_RANK), ("begin", HS_BEGIN),
                ("finish", HS_FINISH), ("search", HS_SEARCH), ("mid_ranks", HS_MID_RANKS), ("mid", HS_MID),
                ("pre_ranks", HS_PRE_RANKS), ("pre", HS_PRE), ("begin_rest", HS_BEGIN_REST)]

_lib = None


def lib_path() -> str:
    return _build.LIB


def load():
    """Load (building first if the sources are newer and hipcc is available)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP runtimes in one process do not
    # share the device.  Import torch FIRST so that the loader resolves our NEEDED libamdhip64.so.7
    # to the copy that is already mapped (same SONAME) and torch tensors and our kernels share it.
    import torch  # noqa: F401
    path = os.environ.get("DHR_HIP_LIB") or _build.LIB      # the override is a tuning aid (alternative builds side by side)
    import shutil
    have_hipcc = shutil.which("hipcc") is not None or os.path.exists("/opt/rocm/bin/hipcc")
    if not os.path.exists(path) or (path == _build.LIB and have_hipcc and _build._stale()):
        # a library older than its sources would be loaded silently -- and a changed struct layout corrupts memory through
        # ctypes -- so rebuild whenever a compiler is at hand (build() is a no-op when nothing is stale)
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001
            raise DhrError(f"libdhr_hip.so is missing or older than its sources and hipcc failed: {e}") from e
    lib = C.CDLL(path)
    lib.dhr_version.restype = C.c_int
    lib.dhr_abi_sizes.argtypes = [C.POINTER(C.c_int32)]
    sizes = (C.c_int32 * 4)()
    lib.dhr_abi_sizes(sizes)
    want = [C.sizeof(IndexDesc), C.sizeof(QueryBatch), C.sizeof(SearchStats), C.sizeof(FileInfo)]
    if list(sizes) != want:
        raise DhrError(f"libdhr_hip.so struct sizes {list(sizes)} differ from the binding's {want}: stale library, rebuild it")
    lib.dhr_abi_size.argtypes = [C.c_int32]
    lib.dhr_abi_size.restype = C.c_int32
    if lib.dhr_abi_size(ABI_HOST_SHARD) != C.sizeof(HostShard):
        raise DhrError(f"libdhr_hip.so: sizeof(dhr_host_shard) = {lib.dhr_abi_size(ABI_HOST_SHARD)}, the binding's {C.sizeof(HostShard)}: stale library, rebuild it")
    lib.dhr_debug_fail_alloc.argtypes = [C.c_int64]
    lib.dhr_debug_fail_alloc.restype = C.c_int64
    lib.dhr_last_error.restype = C.c_char_p
    lib.dhr_index_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]
    lib.dhr_index_destroy.argtypes = [C.c_void_p]
    lib.dhr_index_destroy.restype = None
    lib.dhr_index_set_param.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    lib.dhr_index_device_bytes.argtypes = [C.c_void_p]
    lib.dhr_index_device_bytes.restype = C.c_int64
    lib.dhr_search.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dhr_search_rerank.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.POINTER(QueryBatch), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_int32, C.c_void_p]
    lib.dhr_index_save.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
    lib.dhr_index_file_info.argtypes = [C.c_char_p, C.POINTER(FileInfo)]
    lib.dhr_index_load.argtypes = [C.c_char_p, C.c_int32, C.c_int64, C.POINTER(C.c_void_p)]
    lib.dhr_densify.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
    lib.dhr_pq_train.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                 C.POINTER(C.c_double), C.c_void_p]
    lib.dhr_pq_encode.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dhr_pq_train_nbits.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p,
                                       C.POINTER(C.c_double), C.c_void_p]
    lib.dhr_pq_encode_nbits.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dhr_pq_decode_nbits.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.dhr_write_trec.argtypes = [C.c_char_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_int64, C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    lib.dhr_format_float.argtypes = [C.c_double, C.c_char_p, C.c_int32]
    lib.dhr_pq_decode.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.dhr_score_rows.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dhr_merge_topk.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    lib.dhr_merge_topk_host.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.dhr_merge_topk_lists.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    lib.dhr_merge_topk_lists_host.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                              C.c_void_p]
    lib.dhr_comm_unique_id.argtypes = [C.c_void_p, C.c_int32]
    lib.dhr_comm_create.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.dhr_comm_wrap.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.dhr_comm_create_callback.argtypes = [C.c_int32, C.c_int32, C.c_int32, ALLGATHER_FN, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.dhr_search_sharded_host.argtypes = [C.POINTER(HostShard), C.c_int32, C.c_int32, ALLGATHER_FN, C.c_void_p, C.POINTER(QueryBatch), C.c_int32,
                                            C.c_void_p, C.c_void_p]
    lib.dhr_comm_destroy.argtypes = [C.c_void_p]
    lib.dhr_comm_destroy.restype = None
    lib.dhr_comm_abort.argtypes = [C.c_void_p]
    lib.dhr_comm_abort.restype = None
    lib.dhr_comm_info.argtypes = [C.c_void_p, C.c_int32]
    lib.dhr_search_sharded.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dhr_search_sharded_local.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(QueryBatch), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dhr_pq_create.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.dhr_pq_destroy.argtypes = [C.c_void_p]
    lib.dhr_pq_destroy.restype = None
    lib.dhr_pq_device_bytes.argtypes = [C.c_void_p]
    lib.dhr_pq_device_bytes.restype = C.c_int64
    lib.dhr_pq_search.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dhr_pq_adc_scores.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.dhr_pq_last_scan.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.dhr_get_stats.argtypes = [C.c_void_p, C.POINTER(SearchStats)]
    lib.dhr_set_option.argtypes = [C.c_int32, C.c_int64]
    lib.dhr_index_get_info.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
    lib.dhr_debug_bound_scores.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    lib.dhr_debug_query_margins.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_void_p, C.c_void_p]
    lib.dhr_debug_gemm_time.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.c_void_p]
    if hasattr(lib, "dhr_debug_seq_to_tile"):      # (A/B libraries built from older sources lack it)
        lib.dhr_debug_seq_to_tile.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        lib.dhr_debug_seq_to_tile.restype = None
    lib.dhr_search_sample_rank.argtypes = [C.c_void_p, C.c_int32]
    lib.dhr_search_sample_rank.restype = C.c_int32
    lib.dhr_search_union_rank.argtypes = [C.c_void_p, C.c_int32]
    lib.dhr_search_union_rank.restype = C.c_int32
    lib.dhr_search_begin.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_void_p, C.c_void_p]
    lib.dhr_search_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.dhr_search_mid_ranks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.dhr_search_mid_ranks.restype = C.c_int32
    lib.dhr_search_mid.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.dhr_search_pre_ranks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.dhr_search_pre_ranks.restype = C.c_int32
    lib.dhr_search_pre.argtypes = [C.c_void_p, C.POINTER(QueryBatch), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.dhr_search_begin_rest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != DHR_OK:
        raise DhrError(f"{what} failed ({rc}): {load().dhr_last_error().decode()}", status=rc)


_NP_IDX = {np.dtype(np.uint8): IDX_U8, np.dtype(np.int8): IDX_I8, np.dtype(np.int16): IDX_I16}


def idx_code(dtype) -> int:
    """numpy/torch integer dtype of an index array -> dhr_idx_dtype."""
    name = str(dtype).replace("torch.", "")
    try:
        return _NP_IDX[np.dtype(name)]
    except (KeyError, TypeError):
        raise DhrError(f"unsupported index dtype {dtype} (uint8 / int8 / int16)") from None


def _ptr_ld(a):
    """(pointer, leading dimension in elements, mem_kind) of a 2-D numpy array or torch tensor."""
    if isinstance(a, np.ndarray):
        if a.ndim != 2 or a.strides[1] != a.itemsize or a.strides[0] % a.itemsize:
            raise DhrError("arrays must be 2-D with a contiguous last dimension")
        return a.ctypes.data, a.strides[0] // a.itemsize, MEM_HOST
    if a.dim() != 2 or a.stride(1) != 1:
        raise DhrError("tensors must be 2-D with a contiguous last dimension")
    return a.data_ptr(), a.stride(0), (MEM_DEVICE if a.is_cuda else MEM_HOST)


def _val_code(a) -> int:
    name = str(a.dtype).replace("torch.", "")
    if name == "float16":
        return VAL_F16
    if name == "float32":
        return VAL_F32
    raise DhrError(f"unsupported value dtype {a.dtype} (float16 / float32)")


def make_query_batch(value, index):
    """-> (QueryBatch, keepalive).  value [Q,K] fp16|fp32, index [Q,D] or None; numpy (host) or
    torch (host/device) -- both must live in the same memory kind."""
    qb = QueryBatch()
    p, ld, kind = _ptr_ld(value)
    qb.n_queries = int(value.shape[0])
    qb.mem_kind = kind
    qb.value, qb.ld_value, qb.value_dtype = p, ld, _val_code(value)
    if index is not None:
        pi, ldi, kind_i = _ptr_ld(index)
        if kind_i != kind:
            raise DhrError("query value and index must live in the same memory kind")
        qb.index, qb.ld_index, qb.index_dtype = pi, ldi, idx_code(index.dtype)
    else:
        qb.index, qb.ld_index, qb.index_dtype = None, 0, IDX_NONE
    return qb, (value, index)
