"""CPU-only checks of the C-ABI boundary: the library builds/loads, exports every symbol that
include/dhr_hip.h declares, fails loudly without a GPU, and its host-side reduce matches the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dhr_hip.h")).read()
    return sorted(set(re.findall(r"\b(dhr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from dhr_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dhr_hip.h but not exported"
    assert sorted(_lib.EXPORTS) == names
    assert lib.dhr_version() == 105


def test_struct_layouts_match_header():
    from dhr_amd import _lib
    assert C.sizeof(_lib.IndexDesc) == 72
    assert C.sizeof(_lib.QueryBatch) == 48
    assert C.sizeof(_lib.SearchStats) == 128


def test_invalid_arguments_return_codes_not_crashes():
    from dhr_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    d = _lib.IndexDesc()
    assert lib.dhr_index_create(None, C.byref(h)) == -1
    d.n_rows = 0
    assert lib.dhr_index_create(C.byref(d), C.byref(h)) == -1
    assert b"n_rows" in lib.dhr_last_error()
    v = np.zeros((4, 20), np.float16)
    d.n_rows, d.d_dlr, d.d_cls, d.value, d.ld_value = 4, 12, 8, v.ctypes.data, 20
    i = np.zeros((4, 12), np.uint8)
    d.index, d.index_dtype, d.ld_index = i.ctypes.data, _lib.IDX_U8, 12
    d.d_cls = 9000
    assert lib.dhr_index_create(C.byref(d), C.byref(h)) == -1          # ld_value too small for d_dlr + d_cls
    d.d_cls, d.ld_value = 8190, 8202
    assert lib.dhr_index_create(C.byref(d), C.byref(h)) == -2          # more than 8192 columns: legal input the kernels do not cover
    assert b"8192" in lib.dhr_last_error()
    assert lib.dhr_merge_topk_host(0, 1, None, None, 1, None, None) == -1


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dhr_amd import _lib
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    with pytest.raises(_lib.DhrError, match="dhr_index_create failed"):
        GipIndex(np.zeros((16, 64), np.float16))


def test_host_merge_matches_oracle():
    from dhr_amd import _lib
    from oracle import gip_oracle as O
    lib = _lib.load()
    rng = np.random.default_rng(1)
    q, n_in, k = 11, 257, 40
    s = np.round(rng.standard_normal((q, n_in)).astype(np.float32), 1)
    r = rng.permutation(100000)[: q * n_in].reshape(q, n_in).astype(np.int64)
    r[:, ::17] = -1
    s[0, :] = -np.inf                                  # all -inf scores still rank by row
    es, er = O.merge_topk([s], [r], k)
    hs, hr = np.empty((q, k), np.float32), np.empty((q, k), np.int64)
    assert lib.dhr_merge_topk_host(q, n_in, s.ctypes.data, r.ctypes.data, k, hs.ctypes.data, hr.ctypes.data) == 0
    np.testing.assert_array_equal(hs, es)
    np.testing.assert_array_equal(hr, er)
    # k_out larger than the valid entries -> (-inf, -1) padding
    hs2, hr2 = np.empty((q, 300), np.float32), np.empty((q, 300), np.int64)
    assert lib.dhr_merge_topk_host(q, n_in, s.ctypes.data, r.ctypes.data, 300, hs2.ctypes.data, hr2.ctypes.data) == 0
    valid = (r >= 0).sum(1)
    for i in range(q):
        assert np.all(hr2[i, valid[i]:] == -1) and np.all(np.isneginf(hs2[i, valid[i]:]))


def test_host_merge_signed_zero_ties():
    """-0.0 and +0.0 are ONE score (they compare equal in torch.topk and in Python's sorted, merge.result.py:36): the reduces must tie them
    and fall to "row asc" -- until round 6 the order-preserving key put every +0.0 before every -0.0 (found by tools/stress_modes.py)."""
    from dhr_amd import _lib
    from oracle import gip_oracle as O
    lib = _lib.load()
    rng = np.random.default_rng(3)
    q, n_in, k = 5, 64, 48
    s = np.where(rng.random((q, n_in)) < 0.5, np.float32(0.0), np.float32(-0.0)).astype(np.float32)
    s[:, :8] = np.float32(1.5)
    s[:, 8:12] = np.float32(-2.0)
    r = np.stack([rng.permutation(1000)[:n_in] for _ in range(q)]).astype(np.int64)
    es, er = O.merge_topk([s], [r], k)
    hs, hr = np.empty((q, k), np.float32), np.empty((q, k), np.int64)
    assert lib.dhr_merge_topk_host(q, n_in, s.ctypes.data, r.ctypes.data, k, hs.ctypes.data, hr.ctypes.data) == 0
    np.testing.assert_array_equal(hr, er)
    np.testing.assert_array_equal(hs, es)
    # the same entries as 4 sorted lists of 16
    order = np.stack([np.concatenate([c * 16 + np.lexsort((r[i, c * 16:(c + 1) * 16], -s[i, c * 16:(c + 1) * 16].astype(np.float64))) for c in range(4)]) for i in range(q)])
    sl = np.ascontiguousarray(np.take_along_axis(s, order, 1).reshape(q, 4, 16).transpose(1, 0, 2))
    rl = np.ascontiguousarray(np.take_along_axis(r, order, 1).reshape(q, 4, 16).transpose(1, 0, 2))
    assert lib.dhr_merge_topk_lists_host(q, 4, 16, sl.ctypes.data, rl.ctypes.data, k, hs.ctypes.data, hr.ctypes.data) == 0
    np.testing.assert_array_equal(hr, er)
    np.testing.assert_array_equal(hs, es)


def test_host_merge_lists_matches_oracle():
    from dhr_amd import _lib
    from oracle import gip_oracle as O
    lib = _lib.load()
    from tests.util import sorted_lists
    rng = np.random.default_rng(5)
    n_lists, q, ll, k = 5, 9, 60, 100
    s, r = sorted_lists(rng, n_lists, q, ll)
    es, er = O.merge_topk(list(s), list(r), k)
    hs, hr = np.empty((q, k), np.float32), np.empty((q, k), np.int64)
    assert lib.dhr_merge_topk_lists_host(q, n_lists, ll, s.ctypes.data, r.ctypes.data, k, hs.ctypes.data, hr.ctypes.data) == 0
    np.testing.assert_array_equal(hs, es)
    np.testing.assert_array_equal(hr, er)
    # scores only: the k best scores of the union
    hs2 = np.empty((q, 7), np.float32)
    full = np.where(r >= 0, s, -np.inf)
    assert lib.dhr_merge_topk_lists_host(q, n_lists, ll, full.ctypes.data, None, 7, hs2.ctypes.data, None) == 0
    want = -np.sort(-full.transpose(1, 0, 2).reshape(q, -1), axis=1)[:, :7]
    np.testing.assert_array_equal(hs2, want)
    assert lib.dhr_merge_topk_lists_host(0, 1, 1, None, None, 1, None, None) == -1


def test_file_info_struct_and_bad_files(tmp_path):
    """dhr_file_info layout matches the header; a file that is not a device-ready index is rejected without a GPU."""
    import ctypes as C
    from dhr_amd import _lib
    assert C.sizeof(_lib.FileInfo) == 64
    lib = _lib.load()
    info = _lib.FileInfo()
    bad = tmp_path / "not_an_index.bin"
    bad.write_bytes(b"\x80\x04" + b"x" * 8000)                     # e.g. somebody passes the pickle
    assert lib.dhr_index_file_info(str(bad).encode(), C.byref(info)) < 0
    assert b"not a device-ready index file" in lib.dhr_last_error()
    short = tmp_path / "short.bin"
    short.write_bytes(_lib.FILE_MAGIC + b"\0" * 100)
    assert lib.dhr_index_file_info(str(short).encode(), C.byref(info)) < 0
    assert lib.dhr_index_file_info(str(tmp_path / "missing").encode(), C.byref(info)) < 0
    h = C.c_void_p()
    assert lib.dhr_index_load(str(bad).encode(), 0, -1, C.byref(h)) < 0 and not h.value


def test_division_free_tile_map_equals_integer_division():
    """The bound-GEMM kernels map a launch position to a corpus tile with reciprocal multiplications (dhr_internal.h:
    seq_to_tile_fast); the library's host build of the same function must agree with plain integer division for every
    map mode, at the extremes of the operand range (tile counts below 2^24) and on random inputs."""
    import math
    import random
    from dhr_amd import _lib
    lib = _lib.load()
    out = (C.c_int64 * 2)()
    rng = random.Random(7)

    def check(seq, mode, period, head, mul, n):
        lib.dhr_debug_seq_to_tile(seq, mode, period, head, mul, n, out)
        assert out[0] == out[1], (seq, mode, period, head, mul, n, out[0], out[1])
    for n in (1, 2, 31, 64, 1000, 34275, 65521, (1 << 24) - 1, (1 << 24) - 3):
        mul = int(0.6180339887 * n) | 1
        while math.gcd(mul, n) != 1:
            mul += 2
        for period in (2, 3, 4, 8, 16, 32, 33, 64, 256):
            seqs = {0, 1, n - 1, n // 2, max(0, n - 2)} | {rng.randrange(n) for _ in range(40)}
            for seq in seqs:
                for mode in (0, 1, 2, 3):
                    check(seq, mode, period, rng.randrange(0, 9), mul, n)
    # quotients that are exact multiples (the one place where a truncated estimate can come out one short)
    for n in (31, 4096, 999983, (1 << 24) - 1):
        for m in (0, 1, 2, 12345, n - 1):
            check(m, 3, 32, 4, n - 1 if n > 2 else 1, n)      # (m * (n - 1)) % n
            check(m * 31 % max(n, 1), 2, 32, 0, 1, n)
