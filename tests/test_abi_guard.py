"""The exception barrier of the C ABI ("no C++ exceptions or exit() cross the boundary"; dhr_amd/csrc/abi_guard.h).

CPU part (-m "not gpu"): (i) every extern "C" definition in dhr_amd/csrc is a function-try-block; (ii) real allocation failures, injected
with the library-private operator new (dhr_debug_fail_alloc / DHR_TEST_FAIL_ALLOC), come back from host-only entry points as DHR_ERR_NOMEM
with a message -- every single host allocation of the call in turn --, from dhr_index_create / dhr_search at their entry checkpoint, and
from dhr_search_sharded_host in the middle of the sharded control flow; (iii) the process' own allocator is not interposed.
GPU part: the same sweep through dhr_index_create and dhr_search with a device behind them (tests/test_gpu_parity.py)."""
import ctypes as C
import glob
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dhr_amd import _lib  # noqa: E402

# extern "C" definitions that need no try-block: their bodies cannot throw (fixed buffers, snprintf, atomics) -- abi.cpp
NOTHROW = {"dhr_version", "dhr_last_error", "dhr_set_error_message", "dhr_abi_sizes", "dhr_abi_size", "dhr_debug_fail_alloc"}


def _extern_c_definitions():
    out = []
    for path in sorted(glob.glob(os.path.join(ROOT, "dhr_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "dhr_amd", "csrc", "*.cpp"))):
        src = open(path).read()
        for m in re.finditer(r'^extern "C" [^\n(]*?(\w+)\(', src, re.M):
            # skip to the end of the parameter list
            i, depth = m.end() - 1, 0
            while True:
                depth += src[i] == "("
                depth -= src[i] == ")"
                i += 1
                if depth == 0:
                    break
            rest = src[i:i + 40].lstrip()
            if rest.startswith(";"):
                continue                      # a declaration
            out.append((os.path.basename(path), m.group(1), rest))
    return out


def test_every_extern_c_definition_is_a_function_try_block():
    defs = _extern_c_definitions()
    assert len(defs) >= 70
    names = {n for _, n, _ in defs}
    hdr = open(os.path.join(ROOT, "include", "dhr_hip.h")).read()
    declared = set(re.findall(r"\b(dhr_[a-z0-9_]+)\s*\(", hdr)) - {"dhr_allgather_fn"}
    assert declared <= names | {"dhr_host_shard"}, declared - names
    for path, name, rest in defs:
        if name in NOTHROW:
            continue
        assert rest.startswith("try"), f"{path}: extern \"C\" {name} is not a function-try-block (abi_guard.h)"
    # ... and each of them ends in one of the three handlers
    for path in glob.glob(os.path.join(ROOT, "dhr_amd", "csrc", "*.hip")):
        src = open(path).read()
        n_try = len(re.findall(r'^extern "C" [^;{]*?\)\s*try\s*\{', src, re.M | re.S))
        n_catch = len(re.findall(r"\} DHR_CATCH_(STATUS|VOID|VALUE\()", src))
        assert n_try == n_catch, (path, n_try, n_catch)


def _sweep(lib, call, expect_ok):
    """Fail every host allocation of `call` in turn: each failure must come back as DHR_ERR_NOMEM with a message (no crash, no exception
    through ctypes); afterwards the call works."""
    a0 = lib.dhr_debug_fail_alloc(0)
    assert call() == expect_ok
    n_alloc = lib.dhr_debug_fail_alloc(0) - a0
    assert n_alloc >= 1
    for i in range(1, n_alloc + 1):
        lib.dhr_debug_fail_alloc(i)
        rc = call()
        lib.dhr_debug_fail_alloc(0)
        assert rc == _lib.ERR_NOMEM, (i, n_alloc, rc, lib.dhr_last_error())
        assert b"memory" in lib.dhr_last_error()
    assert call() == expect_ok
    return n_alloc


def test_allocation_failures_in_host_entry_points(tmp_path):
    lib = _lib.load()
    rng = np.random.default_rng(3)
    # the shard reduce on host pointers
    Q, n, k = 4, 100, 10
    s = rng.random((Q, n), np.float32)
    r = np.arange(Q * n, dtype=np.int64).reshape(Q, n)
    os_, orow = np.zeros((Q, k), np.float32), np.zeros((Q, k), np.int64)
    _sweep(lib, lambda: lib.dhr_merge_topk_host(Q, n, s.ctypes.data, r.ctypes.data, k, os_.ctypes.data, orow.ctypes.data), 0)
    want = np.sort(s, axis=1)[:, ::-1][:, :k]
    np.testing.assert_array_equal(os_, want)
    L = n // 4
    sl = np.ascontiguousarray(-np.sort(-s.reshape(Q, 4, L), axis=2).transpose(1, 0, 2))
    rl = np.ascontiguousarray(np.arange(4 * Q * L, dtype=np.int64).reshape(4, Q, L))
    _sweep(lib, lambda: lib.dhr_merge_topk_lists_host(Q, 4, L, sl.ctypes.data, rl.ctypes.data, k, os_.ctypes.data, orow.ctypes.data), 0)
    np.testing.assert_array_equal(os_, want)
    # the TREC writer: worker threads allocate too -- an exception must not leave a thread function
    qids = [b"q%d" % i for i in range(Q)]
    dids = [b"d%d" % i for i in range(Q * n)]
    def blob(ids):
        off = np.zeros(len(ids) + 1, np.int64)
        off[1:] = np.cumsum([len(x) + 1 for x in ids])
        return b"\n".join(ids) + b"\n", off
    qb, qo = blob(qids)
    db, do = blob(dids)
    path = str(tmp_path / "run.trec").encode()
    lines = C.c_int64(0)
    rows = np.ascontiguousarray(r[:, :k])
    sc = np.ascontiguousarray(s[:, :k])
    for threads in (1, 3):
        n_alloc = _sweep(lib, lambda: lib.dhr_write_trec(path, 0, Q, k, qb, qo.ctypes.data, db, do.ctypes.data, Q * n, rows.ctypes.data, 0,
                                                         sc.ctypes.data, b"run", 1, threads, C.byref(lines)), 0)
        assert n_alloc >= threads and lines.value == Q * k
    assert open(path).read().count("\n") == Q * k


def test_entry_checkpoints_without_a_device():
    """dhr_index_create / dhr_search / dhr_search_rerank / dhr_score_rows count as an allocation at their entry (before they touch the device), so
    the injection reaches them on a host without a GPU too."""
    lib = _lib.load()
    cv = np.zeros((8, 64), np.float16)
    d = _lib.IndexDesc(0, _lib.MEM_HOST, 8, 0, 64, cv.ctypes.data, 64, None, _lib.IDX_NONE, 0, 0, 0)
    h = C.c_void_p()
    lib.dhr_debug_fail_alloc(1)
    rc = lib.dhr_index_create(C.byref(d), C.byref(h))
    lib.dhr_debug_fail_alloc(0)
    assert rc == _lib.ERR_NOMEM and not h.value and b"memory" in lib.dhr_last_error()
    # (dhr_search and friends validate their handle first; a NULL handle is DHR_ERR_INVALID, not a crash)
    qb, keep = _lib.make_query_batch(np.zeros((2, 64), np.float32), None)
    out_s, out_r = np.zeros((2, 1), np.float32), np.zeros((2, 1), np.int64)
    assert lib.dhr_search(None, C.byref(qb), 1, out_s.ctypes.data, out_r.ctypes.data, _lib.MEM_HOST, None) == _lib.ERR_INVALID
    del keep


def test_env_variable_arms_the_hook():
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from dhr_amd import _lib; lib = _lib.load();"
            "s = np.zeros((1, 4), np.float32); r = np.arange(4, dtype=np.int64).reshape(1, 4); o = np.zeros((1, 2), np.float32); p = np.zeros((1, 2), np.int64);"
            "rc = lib.dhr_merge_topk_host(1, 4, s.ctypes.data, r.ctypes.data, 2, o.ctypes.data, p.ctypes.data);"
            "rc2 = lib.dhr_merge_topk_host(1, 4, s.ctypes.data, r.ctypes.data, 2, o.ctypes.data, p.ctypes.data);"
            "print(rc, rc2, lib.dhr_last_error().decode())") % ROOT
    env = dict(os.environ, DHR_TEST_FAIL_ALLOC="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) == _lib.ERR_NOMEM and int(rc2) == 0 and "memory" in msg


def test_hook_is_private_to_the_library():
    """The replacement operator new is LOCAL to libdhr_hip.so (libdhr.map): another C++ library of the process keeps the global one."""
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True)
    if nm.returncode != 0:
        pytest.skip("nm not available")
    exported = {ln.split()[-1] for ln in nm.stdout.splitlines() if ln.strip()}
    assert not ({"_Znwm", "_Znam", "_ZdlPv", "_ZdaPv", "_ZdlPvm", "_ZdaPvm"} & exported)
    lib = _lib.load()
    import torch
    lib.dhr_debug_fail_alloc(1)
    t = torch.arange(1000).tolist()                     # allocations of other libraries do not trip the armed counter
    assert lib.dhr_debug_fail_alloc(0) >= 0 and len(t) == 1000
    s = np.zeros((1, 4), np.float32); r = np.arange(4, dtype=np.int64).reshape(1, 4); o = np.zeros((1, 2), np.float32); p = np.zeros((1, 2), np.int64)
    lib.dhr_debug_fail_alloc(1)
    assert lib.dhr_merge_topk_host(1, 4, s.ctypes.data, r.ctypes.data, 2, o.ctypes.data, p.ctypes.data) == _lib.ERR_NOMEM      # ... the library's own does
    lib.dhr_debug_fail_alloc(0)


class _Shard:
    """Oracle-free host shard for dhr_search_sharded_host with world = 1 (dense scores in float64)."""

    def __init__(self, cv):
        self.cv = cv.astype(np.float64)

    def union_rank(self, k):
        return 12

    def sample_rank(self, k, share):
        return 12

    def _scores(self, q):
        return np.asarray(q, np.float64) @ self.cv.T

    def search_begin(self, q, qi, k, share):
        self.q, self.k = np.array(q), k
        return (-np.sort(-self._scores(q)[:, ::4], axis=1)[:, :12]).astype(np.float32)

    def search_finish(self, tau):
        s = self._scores(self.q)
        k, nq = self.k, s.shape[0]
        sc = np.full((nq, k), -np.inf, np.float32); rows = np.full((nq, k), -1, np.int64); cnt = np.zeros(nq, np.int32)
        for i in range(nq):
            keep = np.nonzero(s[i].astype(np.float32) >= float(tau[i]))[0]
            keep = keep[np.lexsort((keep, -s[i][keep]))][:k]
            sc[i, : len(keep)] = s[i][keep]; rows[i, : len(keep)] = keep; cnt[i] = len(keep)
        return sc, rows, cnt

    def search(self, q, qi, k):
        s = self._scores(np.asarray(q))
        order = np.argsort(-s, axis=1, kind="stable")[:, :k]
        return np.take_along_axis(s, order, 1).astype(np.float32), order


def test_allocation_failures_inside_the_sharded_control_flow():
    """dhr_search_sharded_host: std::vector after std::vector between the shard callbacks and the gathers -- every one of them fails in turn."""
    from dhr_amd import dist as D
    lib = _lib.load()
    rng = np.random.default_rng(11)
    cv = rng.standard_normal((600, 16)).astype(np.float32)
    q = rng.standard_normal((3, 16)).astype(np.float32)
    shard = _Shard(cv)
    a0 = lib.dhr_debug_fail_alloc(0)
    ms, mr = D.sharded_search_host(shard, q, None, 20)
    n_alloc = lib.dhr_debug_fail_alloc(0) - a0
    assert n_alloc >= 10
    full = q.astype(np.float64) @ cv.astype(np.float64).T
    for i in range(3):
        assert set(mr[i].tolist()) == set(np.argsort(-full[i], kind="stable")[:20].tolist())
    for i in range(1, n_alloc + 1):
        lib.dhr_debug_fail_alloc(i)
        try:
            with pytest.raises(_lib.DhrError) as ei:
                D.sharded_search_host(shard, q, None, 20)
        finally:
            lib.dhr_debug_fail_alloc(0)
        assert ei.value.status == _lib.ERR_NOMEM, (i, ei.value)
    ms2, mr2 = D.sharded_search_host(shard, q, None, 20)
    np.testing.assert_array_equal(mr, mr2)


def test_host_shard_struct_size_is_checked():
    lib = _lib.load()
    hs = _lib.HostShard()
    hs.struct_size = C.sizeof(_lib.HostShard) - 24            # a caller built against the header before pre_ranks / pre / begin_rest
    qb, keep = _lib.make_query_batch(np.zeros((2, 16), np.float32), None)
    out_s, out_r = np.zeros((2, 1), np.float32), np.zeros((2, 1), np.int64)
    rc = lib.dhr_search_sharded_host(C.byref(hs), 1, 0, _lib.ALLGATHER_FN(lambda *_a: 1), None, C.byref(qb), 1, out_s.ctypes.data, out_r.ctypes.data)
    assert rc == _lib.ERR_INVALID and b"struct_size" in lib.dhr_last_error()
    assert lib.dhr_abi_size(_lib.ABI_HOST_SHARD) == C.sizeof(_lib.HostShard)
    assert lib.dhr_abi_size(99) == _lib.ERR_INVALID
    del keep
