"""The oracle (oracle/gip_oracle.py) against outputs of the reference itself (tests/golden, made by
tests/golden/make_golden.py in the build container).  CPU only."""
import os
import tempfile

import numpy as np
import pytest

from oracle import gip_oracle as O
from tests.util import case_args, dump_pickle, parse_trec

FN_CASES = ["F1_bm25_brute", "F1b_mix8_brute", "F3_hyb_brute_k100", "F3_hyb_brute_k1000", "F4_hyb128_brute",
            "F5_hyb_lamda05", "F5_hyb_lamda03", "F6_hyb_theta03_rerank", "F6_hyb_theta03_norerank",
            "F6_hyb_ip_rerank", "F6_hyb_ip_norerank", "F7_hyb_shard0of3", "F7_hyb_shard1of3", "F7_hyb_shard2of3"]


def _prepared(golden, info):
    d = golden.inputs(info["inputs"])
    q, qi = O.prepare_queries(d["qv"], d["qi"], info.get("emb_dim", 768), info.get("lamda", 1.0))
    lo, hi = info.get("row_lo", 0), info.get("row_hi", d["cv"].shape[0])
    c = d["cv"][lo:hi].astype(np.float32)
    ci = None if d["ci"] is None else d["ci"][lo:hi]
    return d, q, qi, c, ci


def _compare(ref_rows, ref_scores, rows, scores, exact_of):
    """Reference vs oracle for one query: same score multiset (fp32 noise), same set of rows except
    inside exact ties / the fp32-noise band at the boundary."""
    assert len(rows) == len(ref_rows)
    np.testing.assert_allclose(np.sort(scores), np.sort(ref_scores), rtol=2e-6, atol=2e-6)
    if set(rows) != set(ref_rows):
        ex = exact_of(np.array(sorted(set(rows) ^ set(ref_rows))))
        kth = min(scores)
        assert np.all(np.abs(ex - kth) <= 1e-5 * max(1.0, abs(kth))), "set differs outside the tie band"


@pytest.mark.parametrize("case", FN_CASES)
def test_gip_retrieval_matches_reference(golden, case):
    info, ref_rows, ref_scores = golden.case(case)
    d, q, qi, c, ci = _prepared(golden, info)
    res, sc = O.GIP_retrieval(list(d["qids"]), q, qi, c, ci, case_args(info))
    two_stage = info.get("theta", 0) > 0 and not info.get("brute_force", False)
    for i, qid in enumerate(d["qids"]):
        if two_stage:
            # stage scores are approximations; only the reported score multiset/sets are compared
            exact_of = lambda rows_, i=i: np.zeros(len(rows_)) + min(sc[qid])  # noqa: E731
            if info.get("rerank"):
                exact_of = lambda rows_, i=i: O.gip_scores_f64(q[i], qi[i], c[rows_], ci[rows_])  # noqa: E731
        else:
            exact_of = lambda rows_, i=i: O.gip_scores_f64(q[i], qi[i], c[rows_], ci[rows_])  # noqa: E731
        _compare(ref_rows[i].tolist(), ref_scores[i], res[qid], np.array(sc[qid], np.float32), exact_of)
        # the oracle's own fp32 scores agree with its exact f64 scores
        if not two_stage:
            ex = O.gip_scores_f64(q[i], qi[i], c, ci)
            O.check_topk(res[qid], sc[qid], ex, info["topk"])
            O.check_topk(ref_rows[i], ref_scores[i], ex, info["topk"])      # the reference passes its own contract


def test_ip_retrieval_matches_reference(golden):
    info, ref_rows, ref_scores = golden.case("F2_dense_ip")
    d = golden.inputs("dense")
    q, _ = O.prepare_queries(d["qv"], None, 768, 1.0)
    c = d["cv"].astype(np.float32)
    res, sc = O.IP_retrieval(list(d["qids"]), q, c, case_args(info))
    for i, qid in enumerate(d["qids"]):
        _compare(ref_rows[i].tolist(), ref_scores[i], res[qid], np.array(sc[qid], np.float32),
                 lambda rows_, i=i: O.gip_scores_f64(q[i], None, c[rows_], None))
        assert res[qid] == ref_rows[i].tolist()          # dense random scores: no ties, same order


def test_k_larger_than_n(golden):
    exc = golden.meta["exceptions"]["F8_gip_k_gt_n"]
    assert exc[0] == "RuntimeError"
    d = golden.inputs("hyb")
    q, qi = O.prepare_queries(d["qv"], d["qi"], 768, 1.0)
    info = dict(topk=100, brute_force=True)
    with pytest.raises(RuntimeError, match="out of range"):
        O.GIP_retrieval(list(d["qids"]), q, qi, d["cv"][:64].astype(np.float32), d["ci"][:64], case_args(info))
    info, ref_rows, ref_scores = golden.case("F8_ip_k_gt_n")
    dd = golden.inputs("dense")
    qd, _ = O.prepare_queries(dd["qv"], None, 768, 1.0)
    res, sc = O.IP_retrieval(list(dd["qids"]), qd, dd["cv"][:64].astype(np.float32), case_args(info))
    assert ref_rows.shape[1] == 64
    for i, qid in enumerate(dd["qids"]):
        assert res[qid] == ref_rows[i].tolist()


def _write_main_inputs(golden, tmp):
    d = golden.inputs("hyb")
    mq = golden.inputs("main_queries")
    qp, ip_ = os.path.join(tmp, "q.pt"), os.path.join(tmp, "i.pt")
    dump_pickle(qp, mq["qv"], mq["qi"], [str(x) for x in mq["qids"]])
    dump_pickle(ip_, d["cv"], d["ci"], [str(x) for x in d["docids"]])
    return qp, ip_


def _same_trec(ref_text, got_text):
    ref, got = parse_trec(ref_text), parse_trec(got_text)
    assert list(ref) == list(got)                                 # query order
    for qid in ref:
        r, g = ref[qid], got[qid]
        assert len(r) == len(g)
        assert [x[1] for x in r] == [x[1] for x in g]             # rank numbers incl. the self-match gap
        np.testing.assert_allclose([x[2] for x in r], [x[2] for x in g], rtol=2e-6, atol=2e-6)
        assert len(set(x[0] for x in r) ^ set(x[0] for x in g)) <= 2


@pytest.mark.parametrize("fname,kw", [
    ("golden_main_hyb_brute.trec", dict(brute_force=True, topk=100)),
    ("golden_main_hyb_lamda05.trec", dict(brute_force=True, topk=100, lamda=0.5, run_name="dhr")),
    ("golden_main_hyb_theta_rerank.trec", dict(theta=0.3, rerank=True, agip_topk=512, topk=100)),
    ("golden_main_hyb_shard0.trec", dict(brute_force=True, topk=100, total_shrad=3, shrad=0)),
    ("golden_main_hyb_shard1.trec", dict(brute_force=True, topk=100, total_shrad=3, shrad=1)),
    ("golden_main_hyb_shard2.trec", dict(brute_force=True, topk=100, total_shrad=3, shrad=2)),
])
def test_main_trec_bytes(golden, fname, kw):
    with tempfile.TemporaryDirectory() as tmp:
        qp, ip_ = _write_main_inputs(golden, tmp)
        text = O.run_main(qp, ip_, **kw)
    ref = golden.trec(fname)
    _same_trec(ref, text)
    # self-match filter: query 0 carries the id of doc 5 and retrieves it first -> ranks start at 2
    first = ref.splitlines()[0].split(" ")
    if kw.get("shrad", 0) == 0:
        assert first[3] == "2"
    # fp32 summation order differs between torch and numpy (last-bit differences), so lines are
    # byte-identical only where the fp32 score is; every score token is the repr of an fp32 value
    # widened to double, exactly like the reference's (e.g. 69.52739715576172)
    same = sum(1 for a, b in zip(ref.splitlines(), text.splitlines()) if a == b)
    assert same > 0
    for line in text.splitlines()[:200]:
        tok = line.split(" ")[4]
        assert float(np.float32(float(tok))) == float(tok) and repr(float(tok)) == tok


def test_index_merge_and_dense_main(golden):
    dd = golden.inputs("dense")
    m = golden.meta["index_merge"]
    with tempfile.TemporaryDirectory() as tmp:
        b = m["bounds"]
        for i in range(3):
            dump_pickle(os.path.join(tmp, f"msmarco-passage.split{i:02d}.pt"), dd["cv"][b[i]:b[i + 1]], None,
                        [str(x) for x in dd["docids"][b[i]:b[i + 1]]])
        emb, idx, ids = O.merge_index(tmp, "msmarco-passage", order=m["glob_order"])
        assert idx == 0 and m["idx_value"] == 0               # dense merge stores the int 0 (index.py:40-43)
        assert ids == [str(x) for x in golden.calls["F11_merge_dense.ids"]]
        assert float(emb.astype(np.float64).sum()) == float(golden.calls["F11_merge_dense.checksum"][0])
        ip_ = os.path.join(tmp, "merged.pt")
        dump_pickle(ip_, emb, idx, ids)
        qp = os.path.join(tmp, "q.pt")
        dump_pickle(qp, dd["qv"], None, [str(x) for x in dd["qids"]])
        text = O.run_main(qp, ip_, topk=100)
    _same_trec(golden.trec("golden_main_dense_merged.trec"), text)


def test_merge_results_equals_unsharded(golden):
    """merge.result.py semantics: merging the three reference shard runs reproduces the reference's
    unsharded run (same docids per query, re-numbered ranks)."""
    shards = [golden.trec(f"golden_main_hyb_shard{i}.trec") for i in range(3)]
    merged = parse_trec(O.merge_results(shards, topk=100, run_name="h2oloo"))
    full = parse_trec(golden.trec("golden_main_hyb_brute.trec"))
    for qid in full:
        f_docs = [x[0] for x in full[qid]]
        m_docs = [x[0] for x in merged[qid]][: len(f_docs)]
        assert len(set(f_docs) ^ set(m_docs)) <= 2


def test_densify_oracle_matches_reference_outputs():
    """oracle/densify_oracle.py == the reference's densify (tevatron/DHR/utils.py:5-22) on the recorded cases."""
    import os
    import numpy as np
    import pytest
    from oracle import densify_oracle as DO
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "densify_golden.npz"))
    for name, kw in (("a", dict(dims=768)), ("b", dict(dims=8, remove_dims=3)), ("c", dict(dims=16, remove_dims=0))):
        v, i = DO.densify(g[name + "_in"], **kw)
        np.testing.assert_array_equal(v, g[name + "_val"])
        np.testing.assert_array_equal(i, g[name + "_idx"])
        assert v.dtype == g[name + "_val"].dtype and i.dtype == np.int64
    errs = list(g["errors"])
    with pytest.raises(ValueError) as e:
        DO.densify(np.zeros((2, 3, 4), np.float32), dims=4, remove_dims=0)
    assert str(e.value) == errs[0]
    with pytest.raises(ValueError) as e:
        DO.densify(np.zeros((2, 30), np.float32), dims=7, remove_dims=1)
    assert str(e.value) == errs[1]
    v16, i8 = DO.densify_encoded(g["a_in"])
    assert v16.dtype == np.float16 and i8.dtype == np.uint8


def test_pq_oracle_self_consistency():
    """oracle/pq_oracle.py has no golden vectors (faiss absent: parity unpinned); this only checks that the restatement is
    internally consistent: k-means lowers the quantisation error, codes are nearest centroids, and the table-based ADC score
    equals the inner product with the reconstruction."""
    import numpy as np
    from oracle import pq_oracle as PO
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((3000, 32)) * 0.2 + rng.integers(0, 4, (3000, 1))).astype(np.float16)
    q = rng.standard_normal((5, 32)).astype(np.float32)
    cb0 = PO.train(x, 4, iters=0)
    cb = PO.train(x, 4, iters=10)
    assert cb.shape == (4, 256, 8) and PO.mse(x, cb) < 0.8 * PO.mse(x, cb0)
    codes = PO.encode(x, cb)
    xr = PO.decode(codes, cb)
    for m in range(4):
        sub = x[:50, m * 8:(m + 1) * 8].astype(np.float32)
        d = ((sub[:, None, :] - cb[m][None]) ** 2).sum(2)
        assert (np.abs(d[np.arange(50), codes[:50, m]] - d.min(1)) < 1e-5).all()
    np.testing.assert_allclose(PO.adc_scores(q, codes, cb), q.astype(np.float64) @ xr.T.astype(np.float64), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("case", ["F6_hyb_theta03_rerank", "F6_hyb_theta03_norerank", "F6_hyb_ip_rerank", "F6_hyb_ip_norerank"])
def test_two_stage_rule_admits_the_reference(golden, case):
    """The float64 tie-band rule the GPU tests apply to the theta>0 modes (oracle.check_two_stage / check_topk on the
    stage-1 score) accepts the reference's own recorded lists, and rejects a list with ONE wrong row."""
    info, ref_rows, ref_scores = golden.case(case)
    d, q, qi, c, ci = _prepared(golden, info)
    for i in range(q.shape[0]):
        s1 = O.stage1_scores_f64(q[i], qi[i], c, ci, info["theta"], info.get("IP", False))
        ex = O.gip_scores_f64(q[i], qi[i], c, ci)
        rows = np.asarray(ref_rows[i])
        if info.get("rerank", False):
            rr = rows[np.argsort(-ex[rows], kind="stable")]
            O.check_two_stage(rr, ex[rr], s1, ex, info["agip_topk"], info["topk"])
            # one genuinely wrong row: swap the best row for the best row that is NOT in the list and far below the k-th score
            bad = rr.copy()
            outside = np.setdiff1d(np.nonzero(s1 >= np.sort(s1)[::-1][info["agip_topk"] - 1])[0], rr)
            cand = outside[ex[outside] < ex[rr].min() - 1e-3]
            if len(cand):
                bad[0] = cand[0]
                bad = bad[np.argsort(-ex[bad], kind="stable")]
                with pytest.raises(AssertionError):
                    O.check_two_stage(bad, ex[bad], s1, ex, info["agip_topk"], info["topk"])
        else:
            rr = rows[np.argsort(-s1[rows], kind="stable")]
            np.testing.assert_allclose(np.sort(s1[rr]), np.sort(ref_scores[i].astype(np.float64)), rtol=3e-6, atol=3e-6)
            O.check_topk(rr, s1[rr], s1, info["topk"])
            bad = rr.copy()
            far = np.nonzero(s1 < s1[rr].min() - 1e-3)[0]
            bad[0] = far[0]
            bad = bad[np.argsort(-s1[bad], kind="stable")]
            with pytest.raises(AssertionError):
                O.check_topk(bad, s1[bad], s1, info["topk"])


def test_torch_restatement_equals_numpy_oracle():
    """oracle/gip_oracle_torch.py (the timed cpu_baseline legs of bench.py) returns the numpy oracle's top-k sets."""
    from dhr_amd import synth
    from oracle import gip_oracle_torch as OT
    cv, ci, qv, qi = synth.make_pair(31, 3000, 5, 768, 64)
    c32, q32 = cv.astype(np.float32), qv.astype(np.float32)
    for threads in (1, 2):
        _, rows = OT.gip_loop(q32, qi, c32, ci, 50, threads)
        for i in range(5):
            ex = O.gip_scores_f64(q32[i], qi[i], c32, ci)
            O.check_topk(rows[i][np.argsort(-ex[rows[i]], kind="stable")], np.sort(ex[rows[i]])[::-1], ex, 50)
    dv, _, dq, _ = synth.make_pair(32, 2000, 3, 0, 768, kind="dense")
    _, rows = OT.gip_loop(dq.astype(np.float32), None, dv.astype(np.float32), None, 20, 1)
    for i in range(3):
        ex = O.gip_scores_f64(dq[i].astype(np.float32), None, dv.astype(np.float32), None)
        O.check_topk(rows[i][np.argsort(-ex[rows[i]], kind="stable")], np.sort(ex[rows[i]])[::-1], ex, 20)
