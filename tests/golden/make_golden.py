#!/usr/bin/env python
"""Generate the golden fixtures by RUNNING THE REFERENCE ITSELF (build container only).

The reference (/root/reference, read-only, never copied) is imported with three stubs for
packages the image lacks (pickle5 -> pickle, faiss -> empty module, progressbar -> empty module);
its GIP_retrieval / IP_retrieval / main() / index.main() are executed on small seeded inputs and
the inputs + outputs are written next to this script:

    inputs_<name>.npz       corpus/query value (fp16), index arrays, ids
    golden_calls.npz        outputs of the function-level calls (rows, fp32 scores)
    golden_main_*.trec      bytes of result.trec written by the reference main()
    golden_meta.json        case list, flags, recorded exceptions, recorded merge order

Usage (from the repo root):  python tests/golden/make_golden.py
/root/reference does not exist on the GPU box; tests only read the committed files.
"""
import contextlib
import io
import json
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from dhr_amd import synth  # noqa: E402


def import_reference():
    sys.modules.setdefault("pickle5", pickle)
    sys.modules.setdefault("faiss", types.ModuleType("faiss"))
    pb = types.ModuleType("progressbar")
    pb.__all__ = []
    sys.modules.setdefault("progressbar", pb)
    sys.path.insert(0, "/root")
    import reference.retrieval.gip_retrieval as ref_gip
    import reference.retrieval.index as ref_index
    return ref_gip, ref_index


def make_inputs():
    inp = {}
    cv, ci, qv, qi = synth.make_pair(101, 4096, 16, 768, 768, kind="encoder")          # N = 4096, Q = 16 for F2-F4 too (2 048 x 8 / 1 024 x 8 until round 5)
    inp["hyb"] = dict(cv=cv, ci=ci, qv=qv, qi=qi)
    cv, ci, qv, qi = synth.make_pair(102, 4096, 16, 768, 128, kind="encoder")
    inp["hyb128"] = dict(cv=cv, ci=ci, qv=qv, qi=qi)
    cv, ci, qv, qi = synth.make_pair(103, 4096, 16, 768, 0, kind="bm25")
    inp["bm25"] = dict(cv=cv, ci=ci, qv=qv, qi=qi)
    cv, _, qv, _ = synth.make_pair(104, 4096, 16, 0, 768, kind="dense")
    inp["dense"] = dict(cv=cv, qv=qv)
    # int8 corpus index vs int16 query index (densify_corpus.py:70-72 vs densify_query.py:73)
    cv, ci, qv, qi = synth.make_pair(105, 1024, 8, 768, 0, kind="encoder")
    inp["mix8"] = dict(cv=cv, ci=ci.astype(np.int8), qv=qv, qi=qi.astype(np.int16))
    for name, d in inp.items():
        n, q = d["cv"].shape[0], d["qv"].shape[0]
        d["docids"] = np.array([str(7000000 + 3 * i) for i in range(n)])
        d["qids"] = np.array([str(900 + i) for i in range(q)])
    return inp


def as_torch(ref_gip, qv, qi, cv, ci, emb_dim, lamda=1.0):
    """What main() does on the CPU branch (gip_retrieval.py:268-283, 308-315)."""
    import torch
    q = torch.from_numpy(qv.astype(np.float32))
    cls_dim = q.shape[1] - emb_dim
    if cls_dim > 0:
        q[:, -cls_dim:] = lamda * q[:, -cls_dim:]
    c = torch.from_numpy(cv.astype(np.float32))
    tqi = None if qi is None else torch.from_numpy(qi)
    tci = None if ci is None else torch.from_numpy(ci)
    return q, tqi, c, tci


def ns(**kw):
    base = dict(emb_dim=768, theta=0.1, topk=100, agip_topk=512, IP=False, brute_force=False,
                rerank=False)
    base.update(kw)
    return types.SimpleNamespace(**base)


def call_gip(ref_gip, d, args, lamda=1.0, rows=None):
    cv, ci = d["cv"], d["ci"]
    if rows is not None:
        cv, ci = cv[rows[0]:rows[1]], ci[rows[0]:rows[1]]
    q, tqi, c, tci = as_torch(ref_gip, d["qv"], d["qi"], cv, ci, args.emb_dim, lamda)
    qids = list(d["qids"])
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        res, sc = ref_gip.GIP_retrieval(qids, q, tqi, c, tci, args)
    rows_ = np.array([res[q_] for q_ in qids], np.int64)
    scores = np.array([sc[q_] for q_ in qids], np.float32)
    return rows_, scores


def call_ip(ref_gip, d, args):
    q, _, c, _ = as_torch(ref_gip, d["qv"], None, d["cv"], None, 0)
    qids = list(d["qids"])
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        res, sc = ref_gip.IP_retrieval(qids, q, c, args)
    return (np.array([res[q_] for q_ in qids], np.int64), np.array([sc[q_] for q_ in qids], np.float32))


def dump_pickle(path, value, index, ids):
    with open(path, "wb") as f:
        pickle.dump([value, index, list(ids)], f, protocol=4)


def run_ref_main(ref_gip, tmp, qpath, ipath, flags, out_name="result.trec"):
    cwd = os.getcwd()
    os.chdir(tmp)
    argv = sys.argv
    sys.argv = ["gip_retrieval", "--query_emb_path", qpath, "--index_path", ipath] + flags
    try:
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ref_gip.main()
        with open(os.path.join(tmp, out_name), "rb") as f:
            return f.read()
    finally:
        sys.argv = argv
        os.chdir(cwd)


def main():
    ref_gip, ref_index = import_reference()
    inp = make_inputs()
    for name, d in inp.items():
        np.savez_compressed(os.path.join(HERE, f"inputs_{name}.npz"), **d)
    calls, meta = {}, {"cases": {}, "exceptions": {}}

    def rec(case, rows, scores, **info):
        calls[case + ".rows"] = rows
        calls[case + ".scores"] = scores
        meta["cases"][case] = info

    # F1 DLR-only BM25-like, int16 idx, brute force (config 1 shape)
    r, s = call_gip(ref_gip, inp["bm25"], ns(brute_force=True, topk=100))
    rec("F1_bm25_brute", r, s, inputs="bm25", fn="GIP", brute_force=True, topk=100, emb_dim=768)
    # F2 dense-only -> IP_retrieval
    r, s = call_ip(ref_gip, inp["dense"], ns(topk=100))
    rec("F2_dense_ip", r, s, inputs="dense", fn="IP", topk=100)
    # F3 hybrid 768+768 uint8 idx, brute force, k=100 and k=1000
    r, s = call_gip(ref_gip, inp["hyb"], ns(brute_force=True, topk=100))
    rec("F3_hyb_brute_k100", r, s, inputs="hyb", fn="GIP", brute_force=True, topk=100, emb_dim=768)
    r, s = call_gip(ref_gip, inp["hyb"], ns(brute_force=True, topk=1000))
    rec("F3_hyb_brute_k1000", r, s, inputs="hyb", fn="GIP", brute_force=True, topk=1000, emb_dim=768)
    # F4 hybrid 768+128 (the docs' real DeLADE shape)
    r, s = call_gip(ref_gip, inp["hyb128"], ns(brute_force=True, topk=100))
    rec("F4_hyb128_brute", r, s, inputs="hyb128", fn="GIP", brute_force=True, topk=100, emb_dim=768)
    # F5 lamda 0.5
    r, s = call_gip(ref_gip, inp["hyb"], ns(brute_force=True, topk=100), lamda=0.5)
    rec("F5_hyb_lamda05", r, s, inputs="hyb", fn="GIP", brute_force=True, topk=100, emb_dim=768, lamda=0.5)
    r, s = call_gip(ref_gip, inp["hyb"], ns(brute_force=True, topk=100), lamda=0.3)
    rec("F5_hyb_lamda03", r, s, inputs="hyb", fn="GIP", brute_force=True, topk=100, emb_dim=768, lamda=0.3)
    # F6 theta>0 modes
    r, s = call_gip(ref_gip, inp["hyb"], ns(theta=0.3, rerank=True, agip_topk=512, topk=100))
    rec("F6_hyb_theta03_rerank", r, s, inputs="hyb", fn="GIP", theta=0.3, rerank=True, agip_topk=512,
        topk=100, emb_dim=768)
    r, s = call_gip(ref_gip, inp["hyb"], ns(theta=0.3, rerank=False, topk=100))
    rec("F6_hyb_theta03_norerank", r, s, inputs="hyb", fn="GIP", theta=0.3, rerank=False, topk=100,
        emb_dim=768)
    r, s = call_gip(ref_gip, inp["hyb"], ns(theta=0.3, IP=True, rerank=True, agip_topk=512, topk=100))
    rec("F6_hyb_ip_rerank", r, s, inputs="hyb", fn="GIP", theta=0.3, IP=True, rerank=True, agip_topk=512,
        topk=100, emb_dim=768)
    r, s = call_gip(ref_gip, inp["hyb"], ns(theta=0.3, IP=True, rerank=False, topk=100))
    rec("F6_hyb_ip_norerank", r, s, inputs="hyb", fn="GIP", theta=0.3, IP=True, rerank=False, topk=100,
        emb_dim=768)
    # mixed idx dtypes (int8 corpus vs int16 query)
    r, s = call_gip(ref_gip, inp["mix8"], ns(brute_force=True, topk=50))
    rec("F1b_mix8_brute", r, s, inputs="mix8", fn="GIP", brute_force=True, topk=50, emb_dim=768)
    # F7 function-level shards (local row ids within each slice), 3 shards incl. the remainder shard
    n = inp["hyb"]["cv"].shape[0]
    per = n // 3
    for sh in range(3):
        lo, hi = per * sh, (n if sh == 2 else per * (sh + 1))
        r, s = call_gip(ref_gip, inp["hyb"], ns(brute_force=True, topk=100), rows=(lo, hi))
        rec(f"F7_hyb_shard{sh}of3", r, s, inputs="hyb", fn="GIP", brute_force=True, topk=100, emb_dim=768,
            row_lo=lo, row_hi=hi)
    # F8 k >= N: GIP raises, IP silently returns N
    small = {k: (v[:64] if k in ("cv", "ci", "docids") else v) for k, v in inp["hyb"].items()}
    try:
        call_gip(ref_gip, small, ns(brute_force=True, topk=100))
        meta["exceptions"]["F8_gip_k_gt_n"] = None
    except Exception as e:  # noqa: BLE001
        meta["exceptions"]["F8_gip_k_gt_n"] = [type(e).__name__, str(e).splitlines()[0]]
    smalld = {k: (v[:64] if k in ("cv", "docids") else v) for k, v in inp["dense"].items()}
    r, s = call_ip(ref_gip, smalld, ns(topk=100))
    rec("F8_ip_k_gt_n", r, s, inputs="dense", fn="IP", topk=100, n_rows=64)

    # ---- main() end to end (F9 self-match filter, F10 bytes, F7 sharded runs, F2 merged '0' index)
    with tempfile.TemporaryDirectory() as tmp:
        d = inp["hyb"]
        docids = list(d["docids"])
        qids = list(d["qids"])
        qids[0] = docids[5]                 # query id equal to a doc id -> the :340 filter can fire
        qids[1] = docids[77]
        meta["main_qids"] = qids
        qp, ip_ = os.path.join(tmp, "q.pt"), os.path.join(tmp, "i.pt")
        dump_pickle(qp, d["qv"], d["qi"], qids)
        dump_pickle(ip_, d["cv"], d["ci"], docids)
        # make query 0 retrieve the doc that carries its own id: copy doc 5 into query 0
        qv2 = d["qv"].copy(); qi2 = d["qi"].copy()
        qv2[0] = d["cv"][5]; qi2[0] = d["ci"][5]
        qv2[1] = d["cv"][77]; qi2[1] = d["ci"][77]
        dump_pickle(qp, qv2, qi2, qids)
        np.savez_compressed(os.path.join(HERE, "inputs_main_queries.npz"), qv=qv2, qi=qi2, qids=np.array(qids))
        out = run_ref_main(ref_gip, tmp, qp, ip_, ["--brute_force", "--combine_cls", "--topk", "100"])
        open(os.path.join(HERE, "golden_main_hyb_brute.trec"), "wb").write(out)
        out = run_ref_main(ref_gip, tmp, qp, ip_, ["--brute_force", "--topk", "100", "--lamda", "0.5",
                                                  "--run_name", "dhr"])
        open(os.path.join(HERE, "golden_main_hyb_lamda05.trec"), "wb").write(out)
        out = run_ref_main(ref_gip, tmp, qp, ip_, ["--theta", "0.3", "--rerank", "--agip_topk", "512",
                                                  "--topk", "100"])
        open(os.path.join(HERE, "golden_main_hyb_theta_rerank.trec"), "wb").write(out)
        for sh in range(3):
            out = run_ref_main(ref_gip, tmp, qp, ip_, ["--brute_force", "--topk", "100", "--total_shrad", "3",
                                                      "--shrad", str(sh)], out_name=f"result{sh}.trec")
            open(os.path.join(HERE, f"golden_main_hyb_shard{sh}.trec"), "wb").write(out)
        # dense: split files -> index.main() merge (records glob order) -> gip main()
        dd = inp["dense"]
        sp = os.path.join(tmp, "splits"); os.makedirs(sp)
        bounds = [0, 1400, 2800, 4096]
        for i in range(3):
            dump_pickle(os.path.join(sp, f"msmarco-passage.split{i:02d}.pt"), dd["cv"][bounds[i]:bounds[i + 1]],
                        None, list(dd["docids"][bounds[i]:bounds[i + 1]]))
        import glob as _glob
        order = [os.path.basename(p) for p in _glob.glob(os.path.join(sp, "msmarco-passage.split*.pt"))]
        argv = sys.argv
        sys.argv = ["index", "--index_path", sp]
        with contextlib.redirect_stdout(io.StringIO()):
            ref_index.main()
        sys.argv = argv
        with open(os.path.join(sp, "msmarco-passage.index.pt"), "rb") as f:
            m_emb, m_idx, m_ids = pickle.load(f)
        meta["index_merge"] = {"glob_order": order, "idx_value": (m_idx if np.isscalar(m_idx) else "array"),
                               "n": int(m_emb.shape[0]), "first_ids": m_ids[:3], "bounds": bounds}
        calls["F11_merge_dense.ids"] = np.array(m_ids)
        calls["F11_merge_dense.checksum"] = np.array([float(m_emb.astype(np.float64).sum())])
        dqp = os.path.join(tmp, "dq.pt")
        dump_pickle(dqp, dd["qv"], None, list(dd["qids"]))
        out = run_ref_main(ref_gip, tmp, dqp, os.path.join(sp, "msmarco-passage.index.pt"), ["--topk", "100"])
        open(os.path.join(HERE, "golden_main_dense_merged.trec"), "wb").write(out)
        # hybrid split merge (index arrays present)
        sp2 = os.path.join(tmp, "splits2"); os.makedirs(sp2)
        for i in range(3):
            dump_pickle(os.path.join(sp2, f"msmarco-passage.split{i:02d}.pt"), d["cv"][bounds[i]:bounds[i + 1]],
                        d["ci"][bounds[i]:bounds[i + 1]], docids[bounds[i]:bounds[i + 1]])
        order2 = [os.path.basename(p) for p in _glob.glob(os.path.join(sp2, "msmarco-passage.split*.pt"))]
        sys.argv = ["index", "--index_path", sp2]
        with contextlib.redirect_stdout(io.StringIO()):
            ref_index.main()
        sys.argv = argv
        with open(os.path.join(sp2, "msmarco-passage.index.pt"), "rb") as f:
            m_emb, m_idx, m_ids = pickle.load(f)
        meta["index_merge_hyb"] = {"glob_order": order2, "idx_shape": list(m_idx.shape), "n": int(m_emb.shape[0])}
        calls["F11_merge_hyb.ids"] = np.array(m_ids)

    np.savez_compressed(os.path.join(HERE, "golden_calls.npz"), **calls)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(calls), "arrays;", sorted(meta["cases"]))


if __name__ == "__main__":
    main()
