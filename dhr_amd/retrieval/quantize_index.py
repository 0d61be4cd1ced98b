"""Drop-in for castorini/dhr's retrieval/quantize_index.py (:9-42): build the product-quantised first-stage index
of `--PQIP`.  Same flags (including the reference's spelling `--qauntized_dim`).  The reference trains a faiss
`IndexPQ(d, M, nbits, METRIC_INNER_PRODUCT)` and writes it with `faiss.write_index`; faiss is not available to this build, so
the quantiser is the library's own restatement of the algorithm (dhr_pq_train / dhr_pq_encode, HIP).

File format: the output IS a faiss IndexPQ file -- the byte layout faiss' index_write.cpp produces for an IndexPQ
("IxPq" fourcc, index header, ProductQuantizer, codes, search parameters; `write_faiss_indexpq` / `read_faiss_indexpq` below,
restated from the published faiss 1.7 serialisation code) -- so `--faiss_pq_index_path` accepts an index built by the reference's
own quantize_index.py and the reference can read ours.  The layout is restated from the public source, not validated against a
faiss binary here (faiss absent): PARITY WITH FAISS IS UNPINNED -- codebooks differ from faiss' k-means,
tests check the ADC arithmetic exactly against the restated oracle and recall against the exact search.  The earlier pickle
format ({"format": "dhr-pq", ...}) is still read."""
from __future__ import annotations

import argparse
import ctypes as C
import os
import pickle

import numpy as np

from .. import _lib

PQ_FORMAT = "dhr-pq"


def _f16_values(values):
    """The library reads PQ training / encoding input as fp16 rows (the index record's dtype; the C entry points take no dtype argument): anything
    else is converted here -- fp32 that was widened from a record converts back exactly -- instead of being reinterpreted."""
    if isinstance(values, np.ndarray):
        return values if values.dtype == np.float16 else values.astype(np.float16)
    import torch
    return values if values.dtype == torch.float16 else values.half()


def train_and_encode(values, M: int = 64, n_bits: int = 8, iters: int = 25, max_points: int = 65536, device: int = 0):
    """values: fp16 [N, d] numpy array (or a torch CUDA tensor).  -> (codebooks float32 [M, 2^n_bits, d/M], codes uint8 [N,M], mse).
    `--n_bits` as in quantize_index.py:22,29 (faiss.IndexPQ(d, M, nbits)): 1..8; codes are one byte per sub-quantiser here, the
    bit-packed rows of faiss exist in the index file only (pack_codes / unpack_codes)."""
    if not 1 <= int(n_bits) <= 8:
        raise ValueError("--n_bits must be in [1, 8] (faiss' IndexPQ takes up to 24 bits; codes wider than a byte are not built)")
    lib = _lib.load()
    values = _f16_values(values)
    n, d = int(values.shape[0]), int(values.shape[1])
    ksub = 1 << int(n_bits)
    if d % M:
        raise ValueError(f"the vector width {d} is not a multiple of --qauntized_dim {M}")
    p, ld, kind = _lib._ptr_ld(values)
    err = C.c_double()
    if kind == _lib.MEM_HOST:
        cb = np.empty((M, ksub, d // M), np.float32)
        codes = np.empty((n, M), np.uint8)
        pcb, pcodes = cb.ctypes.data, codes.ctypes.data
    else:
        import torch
        cb = torch.empty((M, ksub, d // M), dtype=torch.float32, device=values.device)
        codes = torch.empty((n, M), dtype=torch.uint8, device=values.device)
        pcb, pcodes = cb.data_ptr(), codes.data_ptr()
    _lib.check(lib.dhr_pq_train_nbits(device, kind, p, ld, n, d, M, int(n_bits), iters, max(max_points, ksub), pcb, C.byref(err), None), "dhr_pq_train")
    _lib.check(lib.dhr_pq_encode_nbits(device, kind, p, ld, n, d, M, int(n_bits), pcb, pcodes, None), "dhr_pq_encode")
    return cb, codes, float(err.value)


def encode(values, codebooks, n_bits: int = 8, device: int = 0):
    """Codes of `values` under GIVEN codebooks (dhr_pq_encode: nearest centroid per sub-quantiser): what a shard of a row-sharded
    PQ index does with the corpus-wide codebooks.  values and codebooks live in the same memory kind.  -> codes uint8 [N, M]."""
    lib = _lib.load()
    values = _f16_values(values)
    n, d = int(values.shape[0]), int(values.shape[1])
    M = int(codebooks.shape[0])
    p, ld, kind = _lib._ptr_ld(values)
    if kind == _lib.MEM_HOST:
        cb = np.ascontiguousarray(np.asarray(codebooks), np.float32)
        codes = np.empty((n, M), np.uint8)
        pcb, pcodes = cb.ctypes.data, codes.ctypes.data
    else:
        import torch
        cb = codebooks.contiguous()
        codes = torch.empty((n, M), dtype=torch.uint8, device=values.device)
        pcb, pcodes = cb.data_ptr(), codes.data_ptr()
    _lib.check(lib.dhr_pq_encode_nbits(device, kind, p, ld, n, d, M, int(n_bits), pcb, pcodes, None), "dhr_pq_encode")
    return codes


def decode(codebooks, codes, device: int = 0):
    """-> fp16 [N, d] reconstruction, same memory kind as the inputs."""
    lib = _lib.load()
    M, ksub, dsub = int(codebooks.shape[0]), int(codebooks.shape[1]), int(codebooks.shape[2])
    nbits = ksub.bit_length() - 1
    n, d = int(codes.shape[0]), M * dsub
    if isinstance(codes, np.ndarray):
        out = np.empty((n, d), np.float16)
        codes = np.ascontiguousarray(codes, np.uint8)                     # (locals: the converted copies must outlive the call)
        cb = np.ascontiguousarray(np.asarray(codebooks), np.float32)
        _lib.check(lib.dhr_pq_decode_nbits(device, _lib.MEM_HOST, codes.ctypes.data, n, d, M, nbits, cb.ctypes.data, out.ctypes.data, d, None), "dhr_pq_decode")
        return out
    import torch
    out = torch.empty((n, d), dtype=torch.float16, device=codes.device)
    codes, cb = codes.to(torch.uint8).contiguous(), codebooks.to(torch.float32).contiguous()
    _lib.check(lib.dhr_pq_decode_nbits(device, _lib.MEM_DEVICE, codes.data_ptr(), n, d, M, nbits, cb.data_ptr(), out.data_ptr(), d, None), "dhr_pq_decode")
    return out


FAISS_FOURCC_IXPQ = int.from_bytes(b"IxPq", "little")
METRIC_INNER_PRODUCT, METRIC_L2 = 0, 1


def pack_codes(codes: np.ndarray, nbits: int) -> np.ndarray:
    """[N, M] one code per byte -> faiss' packed code bytes [N, ceil(M * nbits / 8)] (PQEncoderGeneric: LSB first, contiguous)."""
    n, m = codes.shape
    if nbits == 8:
        return np.ascontiguousarray(codes, np.uint8)
    bits = ((codes[:, :, None].astype(np.uint16) >> np.arange(nbits, dtype=np.uint16)) & 1).astype(np.uint8).reshape(n, m * nbits)
    return np.packbits(bits, axis=1, bitorder="little")


def unpack_codes(packed: np.ndarray, m: int, nbits: int) -> np.ndarray:
    if nbits == 8:
        return np.ascontiguousarray(packed[:, :m], np.uint8)
    bits = np.unpackbits(packed, axis=1, bitorder="little")[:, : m * nbits].reshape(packed.shape[0], m, nbits)
    return (bits.astype(np.uint16) << np.arange(nbits, dtype=np.uint16)).sum(2).astype(np.uint8)


def write_faiss_indexpq(path, codebooks, codes, nbits: int = 8, metric: int = METRIC_INNER_PRODUCT):
    """faiss.write_index(IndexPQ) byte layout (faiss/impl/index_write.cpp, 1.7.x):
       u32 fourcc "IxPq" | header: i32 d, i64 ntotal, i64 dummy (1<<20), i64 dummy, u8 is_trained, i32 metric_type
       [, f32 metric_arg if metric_type > 1] | ProductQuantizer: u64 d, u64 M, u64 nbits, vector<float> centroids
       (u64 count, data [M][ksub][dsub]) | vector<u8> codes (u64 count, ntotal * code_size bytes) | i32 search_type (0 = ST_PQ),
       u8 encode_signs, i32 polysemous_ht."""
    cb = np.ascontiguousarray(np.asarray(codebooks), "<f4")
    m, ksub, dsub = cb.shape
    assert ksub == 1 << nbits
    packed = pack_codes(np.asarray(codes, np.uint8), nbits)
    n = packed.shape[0]
    with open(path, "wb") as f:
        f.write(np.uint32(FAISS_FOURCC_IXPQ).tobytes())
        f.write(np.int32(m * dsub).tobytes() + np.int64(n).tobytes() + np.int64(1 << 20).tobytes() + np.int64(1 << 20).tobytes())
        f.write(np.uint8(1).tobytes() + np.int32(metric).tobytes())
        f.write(np.uint64(m * dsub).tobytes() + np.uint64(m).tobytes() + np.uint64(nbits).tobytes())
        f.write(np.uint64(cb.size).tobytes())
        f.write(cb.tobytes())
        f.write(np.uint64(packed.size).tobytes())
        f.write(packed.tobytes())
        f.write(np.int32(0).tobytes() + np.uint8(0).tobytes() + np.int32(0).tobytes())


def read_faiss_indexpq(path):
    """-> {"d", "M", "nbits", "metric", "codebooks": float32 [M, 2^nbits, d/M], "codes": uint8 [N, M] (one code per byte)}."""
    with open(path, "rb") as f:
        buf = memoryview(f.read())
    pos = 0

    def take(dtype, count=1):
        nonlocal pos
        dt = np.dtype(dtype)
        if pos + dt.itemsize * count > len(buf):
            raise ValueError(f"{path}: truncated faiss IndexPQ file")
        a = np.frombuffer(buf, dt, count, pos)
        pos += dt.itemsize * count
        return a
    if int(take("<u4")[0]) != FAISS_FOURCC_IXPQ:
        raise ValueError(f"{path}: not a faiss IndexPQ file (fourcc 'IxPq' expected; other faiss index types are not supported)")
    d, ntotal = int(take("<i4")[0]), int(take("<i8")[0])
    take("<i8", 2)
    _trained, metric = int(take("u1")[0]), int(take("<i4")[0])
    if metric > 1:
        take("<f4")
    pd, m, nbits = (int(x) for x in take("<u8", 3))
    if pd != d or m <= 0 or d % m or not 1 <= nbits <= 8:
        raise ValueError(f"{path}: unsupported ProductQuantizer (d {pd}/{d}, M {m}, nbits {nbits}; nbits <= 8 only)")
    n_cent = int(take("<u8")[0])
    ksub, dsub = 1 << nbits, d // m
    if n_cent != m * ksub * dsub:
        raise ValueError(f"{path}: centroid table of {n_cent} floats, expected {m * ksub * dsub}")
    cb = take("<f4", n_cent).reshape(m, ksub, dsub).copy()
    n_codes = int(take("<u8")[0])
    code_size = (m * nbits + 7) // 8
    if n_codes != ntotal * code_size:
        raise ValueError(f"{path}: {n_codes} code bytes for {ntotal} vectors of {code_size} bytes")
    packed = take("u1", n_codes).reshape(ntotal, code_size)
    if metric != METRIC_INNER_PRODUCT:
        raise ValueError(f"{path}: metric_type {metric}; the reference builds METRIC_INNER_PRODUCT (quantize_index.py:29)")
    return {"format": "faiss-IxPq", "d": d, "M": m, "nbits": nbits, "metric": metric, "codebooks": cb, "codes": unpack_codes(packed, m, nbits)}


class PqIndex:
    """Device-resident PQ index (dhr_pq_*): codes (one byte per sub-quantiser and row) + codebooks; .search = the ADC scan."""

    def __init__(self, codebooks, codes, nbits: int = 8, device: int = 0, row_offset: int = 0):
        self._lib = _lib.load()
        cb = np.ascontiguousarray(np.asarray(codebooks), np.float32) if isinstance(codebooks, np.ndarray) or not hasattr(codebooks, "data_ptr") else codebooks
        self.M, self.ksub, self.dsub = int(cb.shape[0]), int(cb.shape[1]), int(cb.shape[2])
        self.d, self.n, self.device, self.row_offset = self.M * self.dsub, int(codes.shape[0]), int(device), int(row_offset)
        if isinstance(codes, np.ndarray):
            codes = np.ascontiguousarray(codes, np.uint8)
            cb = np.ascontiguousarray(np.asarray(cb), np.float32)
            kind, pc, pb = _lib.MEM_HOST, codes.ctypes.data, cb.ctypes.data
        else:
            codes, cb = codes.contiguous(), cb.contiguous()
            kind, pc, pb = _lib.MEM_DEVICE, codes.data_ptr(), cb.data_ptr()
        h = C.c_void_p()
        _lib.check(self._lib.dhr_pq_create(self.device, kind, self.n, self.d, self.M, int(nbits), pb, pc, self.row_offset, C.byref(h)), "dhr_pq_create")
        self._h = h

    def search(self, q_value, k: int, out_device: bool = False):
        qb, keep = _lib.make_query_batch(q_value, None)
        if out_device:
            import torch
            dev = torch.device("cuda", self.device)
            s = torch.empty((qb.n_queries, k), dtype=torch.float32, device=dev)
            r = torch.empty((qb.n_queries, k), dtype=torch.int64, device=dev)
            _lib.check(self._lib.dhr_pq_search(self._h, C.byref(qb), int(k), s.data_ptr(), r.data_ptr(), _lib.MEM_DEVICE, None), "dhr_pq_search")
            return s, r
        s = np.empty((qb.n_queries, k), np.float32)
        r = np.empty((qb.n_queries, k), np.int64)
        _lib.check(self._lib.dhr_pq_search(self._h, C.byref(qb), int(k), s.ctypes.data, r.ctypes.data, _lib.MEM_HOST, None), "dhr_pq_search")
        return s, r

    def adc_scores(self, q_value, row_lo: int = 0, row_hi: int | None = None):
        import torch
        row_hi = self.n if row_hi is None else row_hi
        qb, keep = _lib.make_query_batch(q_value, None)
        out = torch.empty((qb.n_queries, row_hi - row_lo), dtype=torch.float32, device=torch.device("cuda", self.device))
        _lib.check(self._lib.dhr_pq_adc_scores(self._h, C.byref(qb), int(row_lo), int(row_hi), out.data_ptr(), None), "dhr_pq_adc_scores")
        return out

    def device_bytes(self) -> int:
        return int(self._lib.dhr_pq_device_bytes(self._h))

    def last_scan(self):
        ms, by = C.c_double(), C.c_double()
        _lib.check(self._lib.dhr_pq_last_scan(self._h, C.byref(ms), C.byref(by)), "dhr_pq_last_scan")
        return ms.value, by.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dhr_pq_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def save_pq_pickle(path, codebooks, codes):
    M, _, dsub = codebooks.shape
    with open(path, 'wb') as f:
        pickle.dump({"format": PQ_FORMAT, "version": 1, "d": int(M * dsub), "M": int(M), "nbits": 8,
                     "codebooks": np.asarray(codebooks, np.float32), "codes": np.asarray(codes, np.uint8)}, f, protocol=4)


def save_pq(path, codebooks, codes, nbits: int = 8):
    """What `--output_index_path` receives: a faiss IndexPQ file (see the module docstring)."""
    write_faiss_indexpq(path, np.asarray(codebooks, np.float32), np.asarray(codes, np.uint8), nbits)


def load_pq(path):
    """`--faiss_pq_index_path`: a faiss IndexPQ file (the reference's quantize_index.py output, or ours), or the earlier pickle."""
    with open(path, 'rb') as f:
        head = f.read(4)
    if head == b"IxPq":
        return read_faiss_indexpq(path)
    try:
        with open(path, 'rb') as f:
            obj = pickle.load(f)
    except Exception as e:  # noqa: BLE001
        raise ValueError(f"{path} is neither a faiss IndexPQ file ('IxPq') nor a {PQ_FORMAT} pickle: {e}") from e
    if not isinstance(obj, dict) or obj.get("format") != PQ_FORMAT:
        raise ValueError(f"{path} is neither a faiss IndexPQ file ('IxPq') nor a {PQ_FORMAT} pickle")
    return obj


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--index_path", type=str, required=True)
    parser.add_argument("--output_index_path", type=str, default=None)
    parser.add_argument("--qauntized_dim", type=int, default=64)
    parser.add_argument("--n_bits", type=int, default=8)
    args = parser.parse_args(argv)
    if args.output_index_path is None:
        # assign to index dir (quantize_index.py:16-19; the reference's `index_path` there is an unbound name)
        index_dir = '/'.join(args.index_path.split('/')[:-1])
        args.output_index_path = os.path.join(index_dir, 'pq{}_index'.format(args.qauntized_dim))
    print('Load index ...')
    with open(args.index_path, 'rb') as f:
        corpus_embs, _corpus_arg_idxs, _docids = pickle.load(f)
    corpus_embs = np.ascontiguousarray(np.asarray(corpus_embs), np.float16)
    print('build PQ index...')
    print('train PQ...')
    cb, codes, mse = train_and_encode(corpus_embs, args.qauntized_dim, args.n_bits)
    print('quantisation error (mean squared, per vector): {:.6f}'.format(mse))
    print('write index to {}'.format(args.output_index_path))
    save_pq(args.output_index_path, cb, codes, args.n_bits)
    print('finish')


if __name__ == "__main__":
    main()
