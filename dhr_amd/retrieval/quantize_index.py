"""Drop-in for castorini/dhr's retrieval/quantize_index.py (:9-42): build the product-quantised first-stage index
of `--PQIP`.  Same flags (including the reference's spelling `--qauntized_dim`).  The reference trains and writes a
faiss `IndexPQ(d, M, nbits, METRIC_INNER_PRODUCT)`; faiss is not available to this build, so the quantiser is the
library's own restatement of that algorithm (dhr_pq_train / dhr_pq_encode, HIP) and the file is a pickle

    {"format": "dhr-pq", "version": 1, "d", "M", "nbits", "codebooks": float32 [M,256,d/M], "codes": uint8 [N,M]}

-- NOT a faiss file.  Parity with faiss is unpinned; tests check recall against the exact search."""
from __future__ import annotations

import argparse
import ctypes as C
import os
import pickle

import numpy as np

from .. import _lib

PQ_FORMAT = "dhr-pq"


def train_and_encode(values, M: int = 64, n_bits: int = 8, iters: int = 25, max_points: int = 65536, device: int = 0):
    """values: fp16 [N, d] numpy array (or a torch CUDA tensor).  -> (codebooks float32 [M,256,d/M], codes uint8 [N,M], mse)."""
    if n_bits != 8:
        raise NotImplementedError("only --n_bits 8 (256 centroids per sub-quantiser, the reference's default) is built")
    lib = _lib.load()
    n, d = int(values.shape[0]), int(values.shape[1])
    if d % M:
        raise ValueError(f"the vector width {d} is not a multiple of --qauntized_dim {M}")
    p, ld, kind = _lib._ptr_ld(values)
    err = C.c_double()
    if kind == _lib.MEM_HOST:
        cb = np.empty((M, 256, d // M), np.float32)
        codes = np.empty((n, M), np.uint8)
        pcb, pcodes = cb.ctypes.data, codes.ctypes.data
    else:
        import torch
        cb = torch.empty((M, 256, d // M), dtype=torch.float32, device=values.device)
        codes = torch.empty((n, M), dtype=torch.uint8, device=values.device)
        pcb, pcodes = cb.data_ptr(), codes.data_ptr()
    _lib.check(lib.dhr_pq_train(device, kind, p, ld, n, d, M, iters, max_points, pcb, C.byref(err), None), "dhr_pq_train")
    _lib.check(lib.dhr_pq_encode(device, kind, p, ld, n, d, M, pcb, pcodes, None), "dhr_pq_encode")
    return cb, codes, float(err.value)


def decode(codebooks, codes, device: int = 0):
    """-> fp16 [N, d] reconstruction, same memory kind as the inputs."""
    lib = _lib.load()
    M, dsub = int(codebooks.shape[0]), int(codebooks.shape[2])
    n, d = int(codes.shape[0]), M * dsub
    if isinstance(codes, np.ndarray):
        out = np.empty((n, d), np.float16)
        _lib.check(lib.dhr_pq_decode(device, _lib.MEM_HOST, codes.ctypes.data, n, d, M, np.ascontiguousarray(codebooks, np.float32).ctypes.data,
                                     out.ctypes.data, d, None), "dhr_pq_decode")
        return out
    import torch
    out = torch.empty((n, d), dtype=torch.float16, device=codes.device)
    _lib.check(lib.dhr_pq_decode(device, _lib.MEM_DEVICE, codes.data_ptr(), n, d, M, codebooks.data_ptr(), out.data_ptr(), d, None), "dhr_pq_decode")
    return out


def save_pq(path, codebooks, codes):
    M, _, dsub = codebooks.shape
    with open(path, 'wb') as f:
        pickle.dump({"format": PQ_FORMAT, "version": 1, "d": int(M * dsub), "M": int(M), "nbits": 8,
                     "codebooks": np.asarray(codebooks, np.float32), "codes": np.asarray(codes, np.uint8)}, f, protocol=4)


def load_pq(path):
    with open(path, 'rb') as f:
        obj = pickle.load(f)
    if not isinstance(obj, dict) or obj.get("format") != PQ_FORMAT:
        raise ValueError(f"{path} is not a {PQ_FORMAT} file (faiss index files are not readable without faiss; rebuild it with "
                         "python -m retrieval.quantize_index)")
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
    save_pq(args.output_index_path, cb, codes)
    print('finish')


if __name__ == "__main__":
    main()
