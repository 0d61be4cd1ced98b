"""CPU-only tests of the host side of the drop-in (file formats, shard slicing, TREC writer, index
merge, CLI surface) against the oracle / golden files.  No compute calls."""
import io
import os
import pickle
import tempfile

import numpy as np
import pytest

from oracle import gip_oracle as O
from tests.util import dump_pickle


def test_cli_flags_are_the_reference_flags():
    from dhr_amd.retrieval.gip_retrieval import build_parser
    a = build_parser().parse_args(["--query_emb_path", "q", "--index_path", "i"])
    assert (a.emb_dim, a.theta, a.topk, a.agip_topk, a.batch, a.lamda, a.total_shrad, a.shrad, a.run_name) == \
        (768, 0.1, 1000, 10000, 1, 1, 1, 0, "h2oloo")
    for flag in ("combine_cls", "IP", "PQIP", "brute_force", "use_gpu", "rerank"):
        assert getattr(a, flag) is False
    assert a.faiss_pq_index_path is None


def test_load_queries_and_corpus_shard(golden):
    from dhr_amd.retrieval import gip_retrieval as G
    d = golden.inputs("hyb")
    with tempfile.TemporaryDirectory() as tmp:
        qp, ip_ = os.path.join(tmp, "q.pt"), os.path.join(tmp, "i.pt")
        dump_pickle(qp, d["qv"], d["qi"], list(d["qids"]))
        dump_pickle(ip_, d["cv"], d["ci"], list(d["docids"]))
        q, qi, qids = G.load_queries(qp, 768, 0.3)
        eq, eqi = O.prepare_queries(d["qv"], d["qi"], 768, 0.3)
        np.testing.assert_array_equal(q, eq)
        np.testing.assert_array_equal(qi, eqi)
        assert q.dtype == np.float32 and qids == list(d["qids"])
        for tot in (1, 3, 7):
            for sh in range(tot):
                cv, ci, ids, lo = G.load_corpus_shard(ip_, tot, sh)
                ec, eci, eids = O.prepare_corpus(d["cv"], d["ci"], list(d["docids"]), tot, sh)
                assert cv.dtype == np.float16                    # no fp32 copy on our side
                np.testing.assert_array_equal(cv.astype(np.float32), ec)
                np.testing.assert_array_equal(ci, eci)
                assert ids == eids and lo == O.shard_rows(len(d["docids"]), tot, sh)[0]
        # dense index: None and the merged int 0 both mean "no index array"
        dump_pickle(ip_, d["cv"], 0, list(d["docids"]))
        assert G.load_corpus_shard(ip_)[1] is None
        dump_pickle(ip_, d["cv"], None, list(d["docids"]))
        assert G.load_corpus_shard(ip_)[1] is None
        dump_pickle(qp, d["qv"], None, list(d["qids"]))
        assert G.load_queries(qp, 768, 1.0)[1] is None


def test_trec_writer_matches_oracle_bytes():
    from dhr_amd.retrieval.gip_retrieval import write_trec
    docids = ["d0", "q1", "d2", "d3"]
    results = {"q1": [1, 0, 3], "q9": [2, 3]}
    scores = {"q1": np.array([3.25, 1.1, -0.5], np.float32).tolist(), "q9": np.array([26.177849, 1e-3], np.float32).tolist()}
    buf = io.StringIO()
    write_trec(buf, results, scores, docids, "h2oloo")
    assert buf.getvalue() == "".join(O.trec_lines(results, scores, docids, "h2oloo"))
    assert buf.getvalue().splitlines()[0] == "q1 Q0 d0 2 1.100000023841858 h2oloo"     # self match skipped, gap kept


def test_index_merge_sorted_order(golden):
    from dhr_amd.retrieval import index as I
    d = golden.inputs("hyb")
    dd = golden.inputs("dense")
    b = golden.meta["index_merge"]["bounds"]
    with tempfile.TemporaryDirectory() as tmp:
        for i in (2, 0, 1):       # written out of order: the merge must not depend on directory order
            dump_pickle(os.path.join(tmp, f"msmarco-passage.split{i:02d}.pt"), d["cv"][b[i]:b[i + 1]], d["ci"][b[i]:b[i + 1]],
                        list(d["docids"][b[i]:b[i + 1]]))
        I.main(["--index_path", tmp])
        with open(os.path.join(tmp, "msmarco-passage.index.pt"), "rb") as f:
            emb, idx, ids = pickle.load(f)
        np.testing.assert_array_equal(emb, d["cv"])
        np.testing.assert_array_equal(idx, d["ci"])
        assert ids == list(d["docids"])
        e2, i2, ids2 = O.merge_index(tmp, "msmarco-passage")
        np.testing.assert_array_equal(e2, emb)
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(3):
            dump_pickle(os.path.join(tmp, f"x.split{i:02d}.pt"), dd["cv"][b[i]:b[i + 1]], None, list(dd["docids"][b[i]:b[i + 1]]))
        merged = I.merge_splits(tmp, "x")
        assert merged[1] == 0                                  # index.py:40-43


def test_key_encoding_roundtrip_is_order_preserving():
    """The u64 candidate key used by the select kernel (host twin of dhr_internal.h)."""
    def f32_ordered(f):
        u = np.float32(f).view(np.uint32)
        return (~u) & np.uint32(0xFFFFFFFF) if u & np.uint32(0x80000000) else u | np.uint32(0x80000000)
    vals = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 2.0, 65504.0, np.inf], np.float32)
    keys = [int(f32_ordered(v)) for v in vals]
    assert keys == sorted(keys) and len(set(keys[:3])) == 3


def test_faiss_indexpq_file_round_trip(tmp_path):
    """retrieval/quantize_index.py writes a faiss IndexPQ file (byte layout of faiss' index_write.cpp, restated): header fields at
    the documented offsets, codes packed LSB-first for nbits < 8, and read back bit-identically; other files are refused."""
    import struct
    from dhr_amd.retrieval import quantize_index as QI
    rng = np.random.default_rng(3)
    for nbits, m, dsub, n in ((8, 64, 14, 300), (6, 8, 4, 77), (4, 16, 2, 50)):
        cb = rng.standard_normal((m, 1 << nbits, dsub)).astype(np.float32)
        codes = rng.integers(0, 1 << nbits, (n, m)).astype(np.uint8)
        path = str(tmp_path / f"pq{nbits}")
        QI.write_faiss_indexpq(path, cb, codes, nbits)
        raw = open(path, "rb").read()
        assert raw[:4] == b"IxPq"
        d, ntotal, dummy1, dummy2, trained, metric = struct.unpack_from("<iqqqBi", raw, 4)
        assert (d, ntotal, dummy1, dummy2, trained, metric) == (m * dsub, n, 1 << 20, 1 << 20, 1, 0)
        pd, pm, pnb, ncent = struct.unpack_from("<QQQQ", raw, 4 + 4 + 8 + 8 + 8 + 1 + 4)
        assert (pd, pm, pnb, ncent) == (m * dsub, m, nbits, m * (1 << nbits) * dsub)
        code_size = (m * nbits + 7) // 8
        assert len(raw) == 37 + 32 + 4 * ncent + 8 + n * code_size + 9
        back = QI.load_pq(path)
        assert (back["d"], back["M"], back["nbits"]) == (m * dsub, m, nbits)
        np.testing.assert_array_equal(back["codebooks"], cb)
        np.testing.assert_array_equal(back["codes"], codes)
    np.testing.assert_array_equal(QI.pack_codes(np.array([[1, 2, 3, 0]], np.uint8), 4), np.array([[0x21, 0x03]], np.uint8))   # LSB first
    bad = tmp_path / "bad"
    bad.write_bytes(b"IxFlxxxxxxxxxxxxxxxx")
    with pytest.raises(ValueError):
        QI.load_pq(str(bad))
    trunc = tmp_path / "trunc"
    trunc.write_bytes(open(str(tmp_path / "pq8"), "rb").read()[:500])
    with pytest.raises(ValueError):
        QI.load_pq(str(trunc))
