"""Host mirror of the reference's `densify` (tevatron/DHR/utils.py:5-22) on the fused HIP op `dhr_densify`
(SURVEY section 8f row 4).  Same name, arguments, return values and error messages; works on numpy arrays
(host memory, staged through the device) and on torch tensors (CUDA tensors are processed in place on the device).
There is no CPU implementation here: without the HIP library / a GPU the call raises."""
from __future__ import annotations

import numpy as np

from . import _lib


def _check(lexical_reps, dims, remove_dims):
    if len(lexical_reps.shape) != 2:                                                        # utils.py:11-12
        raise ValueError('Input lexical representation shape should be 2 (batch, vocab), but the input shape is {}'.format(len(lexical_reps.shape)))
    orig_dims = int(lexical_reps.shape[-1])
    if (orig_dims - remove_dims) % dims != 0:                                               # utils.py:14-16
        raise ValueError('Input lexical representation cannot be densified, please fix dims or remove_dims')
    return int(lexical_reps.shape[0]), orig_dims


def _run(lexical_reps, dims, remove_dims, out_value, out_index, device):
    lib = _lib.load()
    p_in, ld_in, kind = _lib._ptr_ld(lexical_reps)
    p_v, ld_v, kind_v = _lib._ptr_ld(out_value)
    p_i, ld_i, kind_i = _lib._ptr_ld(out_index)
    if not (kind == kind_v == kind_i):
        raise _lib.DhrError("densify: input and outputs must live in the same memory kind")
    batch, vocab = int(lexical_reps.shape[0]), int(lexical_reps.shape[1])
    _lib.check(lib.dhr_densify(device, kind, p_in, _lib._val_code(lexical_reps), ld_in, batch, vocab, remove_dims, dims, p_v,
                               _lib._val_code(out_value), ld_v, p_i, _lib.idx_code(out_index.dtype), ld_i, None), "dhr_densify")


def densify(lexical_reps, dims: int = 768, strategy: str = 'stride', remove_dims: int = 570):
    """-> (value_reps [batch, dims] in the input dtype, index_reps [batch, dims] int64), like the reference.
    numpy in -> numpy out; torch in -> torch out (same device)."""
    batch, vocab = _check(lexical_reps, dims, remove_dims)
    n_groups = (vocab - remove_dims) // dims
    if isinstance(lexical_reps, np.ndarray):
        src = np.ascontiguousarray(lexical_reps)
        if src.dtype not in (np.float16, np.float32):
            src = src.astype(np.float32)
        val = np.empty((batch, dims), src.dtype)
        idx = np.empty((batch, dims), np.int16 if n_groups > 256 else np.uint8)
        if batch:
            _run(src, dims, remove_dims, val, idx, 0)
        return val.astype(lexical_reps.dtype, copy=False), idx.astype(np.int64)
    import torch
    src = lexical_reps.detach()
    if src.dtype not in (torch.float16, torch.float32):
        src = src.float()
    src = src.contiguous()
    val = torch.empty((batch, dims), dtype=src.dtype, device=src.device)
    idx = torch.empty((batch, dims), dtype=torch.int16 if n_groups > 256 else torch.uint8, device=src.device)
    if batch:
        _run(src, dims, remove_dims, val, idx, src.device.index or 0 if src.is_cuda else 0)
    return val.to(lexical_reps.dtype), idx.long()


def densify_into(lexical_reps, value_out, index_out, dims: int = 768, remove_dims: int = 570):
    """The encoder driver's use (encode.py:155-170): write the fp16 values into the first `dims` columns of the index
    record's value array (rows of width dims + cls_dim) and the uint8 groups into its index array, in one pass."""
    batch, vocab = _check(lexical_reps, dims, remove_dims)
    if int(value_out.shape[0]) != batch or int(index_out.shape[0]) != batch or int(value_out.shape[1]) < dims or int(index_out.shape[1]) < dims:
        raise ValueError("output arrays do not match the batch / dims")
    if batch:
        dev = 0
        if not isinstance(lexical_reps, np.ndarray) and lexical_reps.is_cuda:
            dev = lexical_reps.device.index or 0
        _run(lexical_reps, dims, remove_dims, value_out, index_out, dev)
    return value_out, index_out
