#!/usr/bin/env python
"""Corrupted device-index files (dhr_index_save's format): random byte flips in the header region, random truncations, random garbage behind a valid magic --
dhr_index_file_info and dhr_index_load answer with a status (or load a file whose damage the header checks cannot see), never with a crash; the intact file
loads and answers as the index it was saved from.  usage: python tools/fuzz_index_file.py [n_mutations] [seed]"""
import ctypes as C
import os
import sys
import tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from dhr_amd import _lib, synth
    from dhr_amd.retrieval import gip_retrieval as G
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp(prefix="dhr_fuzz_")
    stats = dict(refused=0, loaded=0)
    for kind in ("hybrid", "dense"):
        cv, ci, qv, qi = synth.make_pair(5, 3000, 4, 64, 32) if kind == "hybrid" else synth.make_pair(6, 3000, 4, 0, 64, kind="dense")
        q32 = qv.astype(np.float32)
        ix = G.GipIndex(cv, ci)
        s0, r0 = ix.search(q32, qi, 20)
        good = os.path.join(tmp, "good.dhr")
        ix.save(good, docids=["d%d" % i for i in range(3000)])
        ix.close()
        blob = open(good, "rb").read()
        ix2, ids = G.GipIndex.load(good)
        s1, r1 = ix2.search(q32, qi, 20)
        ix2.close()
        assert ids[:3] == ["d0", "d1", "d2"] and np.array_equal(r1, r0) and np.array_equal(s1, s0)
        bad = os.path.join(tmp, "bad.dhr")
        info = _lib.FileInfo()
        for m in range(n_mut):
            b = bytearray(blob)
            how = int(rng.integers(0, 4))
            if how == 0:                                   # a few byte flips in the header region
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, min(len(b), 512)))] = int(rng.integers(0, 256))
            elif how == 1:                                 # a 4- or 8-byte field overwritten with an extreme value
                off = int(rng.integers(8, 256)) & ~3
                val = int(rng.choice([0, 1, 0x7fffffff, 0xffffffff, 0x7fffffffffffffff, 0xffffffffffffffff, len(b), len(b) + 1, 1 << 40]))
                width = int(rng.choice([4, 8]))
                b[off:off + width] = (val & ((1 << (8 * width)) - 1)).to_bytes(width, "little")
            elif how == 2:                                 # truncation
                b = b[: int(rng.integers(0, len(b)))]
            else:                                          # valid magic, garbage behind it
                b = bytearray(_lib.FILE_MAGIC) + bytearray(rng.integers(0, 256, int(rng.integers(0, 4096)), dtype=np.uint8).tobytes())
            with open(bad, "wb") as f:
                f.write(bytes(b))
            rc_info = lib.dhr_index_file_info(bad.encode(), C.byref(info))
            h = C.c_void_p()
            rc = lib.dhr_index_load(bad.encode(), 0, -1, C.byref(h))
            if rc < 0:
                assert not h.value and lib.dhr_last_error(), (kind, m, how)
                stats["refused"] += 1
            else:                                          # damage the header checks cannot see (or none): the handle must work and close
                assert rc_info == 0
                if os.environ.get("DHR_FUZZ_SEARCH", "1") != "0":     # ... and a search on it ends (whatever it returns: the payload may be damaged too)
                    qb, keep = _lib.make_query_batch(q32, qi)
                    hs, hr = np.empty((4, 20), np.float32), np.empty((4, 20), np.int64)
                    print("search on mutation", m, how, flush=True)    # (the last line of the log names the file that crashed the process, should one)
                    rc_s = lib.dhr_search(h, C.byref(qb), 20, hs.ctypes.data, hr.ctypes.data, _lib.MEM_HOST, None)
                    stats["searched_ok" if rc_s == 0 else "search_refused"] = stats.get("searched_ok" if rc_s == 0 else "search_refused", 0) + 1
                    if rc_s == 0 and how != 0 and how != 1:
                        pass
                lib.dhr_index_destroy(h)
                stats["loaded"] += 1
        print(kind, "ok", stats, flush=True)
    print("all mutations answered:", stats)


if __name__ == "__main__":
    main()
