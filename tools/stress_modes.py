#!/usr/bin/env python
"""Randomised differential test of the entry points BESIDE the plain search: the two-stage modes on the device
(dhr_search_rerank: theta > 0 and --IP first stages, gip_retrieval.py:128-156), dhr_score_rows, the index file round trip
(dhr_index_save / dhr_index_load), the fused densify (dhr_densify: ties, all-zero rows, more than 256 groups), the product-quantised first stage (dhr_pq_*: ADC scores and
search, encode / decode, the faiss IndexPQ file; random codebooks, M, sub-vector widths and code widths against oracle/pq_oracle.py), the sharded search in one process (dhr_search_sharded_local over 1 ... 5 row shards of ragged
sizes) and the shard reduces (dhr_merge_topk, dhr_merge_topk_lists, device and host twins) -- random shapes, dtypes, value signs,
bucket counts, k1 / k, both images of the gated half -- against the oracle's float64 scores and its parity rules; and the command line
itself (gip_retrieval.main on random pickles: --lamda, --total_shrad / --shrad, --theta / --IP / --rerank / --brute_force, a query whose own
document is in the corpus, dense-only files) with every line of the run file checked.  Prints the
failing configuration and exits non-zero on the first mismatch.  usage: python tools/stress_modes.py [n_cases] [seed] [only_case]
(only_case: replay -- every other case only draws its random numbers; without a GPU a replayed `merge` case runs the host twins alone)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    import torch
    from dhr_amd import _lib, dist as D
    from dhr_amd.retrieval import gip_retrieval as G
    from oracle import gip_oracle as O
    rng = np.random.default_rng(seed)
    t0 = time.time()
    tmp = tempfile.mkdtemp(prefix="dhr_stress_")
    for case in range(n_cases):
        n = int(rng.choice([300, 1000, 2500, 5000, 20000, 60000]))
        q = int(rng.integers(1, 14))
        d_dlr = int(rng.choice([8, 32, 64, 128, 256, 768]))
        d_cls = int(rng.choice([0, 8, 64, 128, 768]))
        k1 = int(min(n, rng.choice([10, 100, 1000, 3000, 10000, 20000])))
        k = int(min(k1, rng.choice([1, 10, 100, 1000])))
        idx_dtype = rng.choice([np.uint8, np.int8, np.int16])
        n_idx = int(rng.choice([2, 5, 39, 120]))
        neg = bool(rng.random() < 0.25)
        q32 = bool(rng.random() < 0.4)
        nb = int(rng.choice([0, 0, 1, 2]))
        g8 = int(rng.integers(0, 2))
        theta = float(rng.choice([0.05, 0.1, 0.3, 0.6]))
        what = str(rng.choice(["theta", "ip", "score_rows", "file", "local_shards", "merge", "densify", "pq", "cli", "strided", "params"]))
        if os.environ.get("DHR_STRESS_KINDS"):             # e.g. DHR_STRESS_KINDS=params,strided : only these kinds (another random sequence than the full mix)
            what = str(rng.choice(os.environ["DHR_STRESS_KINDS"].split(",")))
        os.environ["DHR_GATED_I8"] = str(g8)
        cfg = dict(case=case, what=what, n=n, q=q, d_dlr=d_dlr, d_cls=d_cls, k1=k1, k=k, idx=np.dtype(idx_dtype).name, n_idx=n_idx,
                   neg=neg, q32=q32, nb=nb, gated_i8=g8, theta=theta)
        K = d_dlr + d_cls
        live = only < 0 or case == only

        def vals(m):
            v = np.abs(rng.standard_normal((m, K))) * 0.5
            v[:, :d_dlr] *= (rng.random((m, d_dlr)) < 0.3)
            if neg:
                v[:, :d_dlr] *= np.where(rng.random((m, d_dlr)) < 0.2, -1.0, 1.0)
            v[:, d_dlr:] = rng.standard_normal((m, d_cls)) * 0.2
            return v
        cv = vals(n).astype(np.float16)
        qv = vals(q)
        qv = (qv * 0.9).astype(np.float32) if q32 else qv.astype(np.float16).astype(np.float32)
        lo = -n_idx // 2 if np.dtype(idx_dtype).kind == "i" else 0
        ci = rng.integers(lo, lo + n_idx, (n, d_dlr)).astype(idx_dtype)
        qi = rng.integers(lo, lo + n_idx, (q, d_dlr)).astype(idx_dtype)
        c32 = cv.astype(np.float32)
        exact = lambda i: O.gip_scores_f64(qv[i], qi[i], c32, ci)       # noqa: E731
        try:
            if what == "merge":
                n_lists, ll = int(rng.integers(1, 10)), int(rng.choice([1, 7, 100, 448, 1000]))
                kk = int(rng.choice([1, 10, 100, 1000]))
                # sorted lists with duplicate scores across lists and ragged padded tails
                sc = np.round(rng.standard_normal((n_lists, q, ll)), int(rng.choice([1, 3, 6]))).astype(np.float32)
                rows = np.stack([rng.permutation(ll * n_lists)[:ll] + 0 for _ in range(n_lists * q)]).reshape(n_lists, q, ll).astype(np.int64)
                for l in range(n_lists):
                    rows[l] += l * 10_000_000           # lists of different shards hold different rows
                    fill = rng.integers(0, ll + 1, q)
                    for i in range(q):
                        o = np.lexsort((rows[l, i], -sc[l, i].astype(np.float64)))
                        sc[l, i], rows[l, i] = sc[l, i][o], rows[l, i][o]
                        sc[l, i, fill[i]:], rows[l, i, fill[i]:] = -np.inf, -1
                cfg.update(n_lists=n_lists, list_len=ll, k_out=kk)
                es, er = O.merge_topk([sc[l] for l in range(n_lists)], [rows[l] for l in range(n_lists)], kk)
                for dev in ("cuda", "cpu"):
                    perm_np = rng.permutation(n_lists * ll)
                    if not live or (dev == "cuda" and not torch.cuda.is_available()):
                        continue
                    cfg["device"] = dev
                    ts, tr = torch.from_numpy(sc).to(dev), torch.from_numpy(rows).to(dev)
                    ms, mr = D.merge_sorted_lists(ts, tr, kk)
                    np.testing.assert_array_equal(mr.cpu().numpy(), er)
                    np.testing.assert_array_equal(ms.cpu().numpy(), es)
                    cs, cr = ts.permute(1, 0, 2).reshape(q, -1), tr.permute(1, 0, 2).reshape(q, -1)
                    perm = torch.from_numpy(perm_np).to(dev)
                    ms, mr = D.merge_topk(cs[:, perm].contiguous(), cr[:, perm].contiguous(), kk)
                    np.testing.assert_array_equal(mr.cpu().numpy(), er)
                    np.testing.assert_array_equal(ms.cpu().numpy(), es)
                    ms, _ = D.merge_sorted_lists(ts, None, kk)
                    np.testing.assert_array_equal(ms.cpu().numpy(), es)
            elif what == "densify":
                from dhr_amd import densify as DZ
                from oracle import densify_oracle as DO
                dims = int(rng.choice([8, 64, 256, 768]))
                groups = int(rng.choice([1, 2, 39, 257, 300]))
                rem = int(rng.choice([0, 5, 570]))
                batch = int(rng.integers(1, 70))
                dt = np.float16 if rng.random() < 0.5 else np.float32
                x = rng.standard_normal((batch, rem + groups * dims)).astype(np.float32)
                mode = int(rng.integers(0, 4))
                if mode == 1:
                    x = np.round(x, 0)                     # many ties: the first maximum wins
                elif mode == 2:
                    x = np.maximum(x, 0) * (rng.random(x.shape) < 0.05)          # sparse, mostly exact zeros (a lexical representation)
                elif mode == 3:
                    x[rng.integers(0, batch)] = 0.0        # an all-zero row
                    x = -np.abs(x)                         # ... and nothing positive elsewhere
                x = x.astype(dt)
                cfg.update(dims=dims, groups=groups, remove_dims=rem, batch=batch, dtype=np.dtype(dt).name, mode=mode)
                if not live:
                    continue
                ev, ei = DO.densify(x, dims, "stride", rem)
                v, i = DZ.densify(x, dims, "stride", rem)
                assert v.dtype == x.dtype and i.dtype == np.int64
                np.testing.assert_array_equal(i, ei)
                np.testing.assert_array_equal(v.view(np.uint16 if dt == np.float16 else np.uint32), ev.view(np.uint16 if dt == np.float16 else np.uint32))
                tv, ti = DZ.densify(torch.from_numpy(x).cuda(), dims, "stride", rem)
                np.testing.assert_array_equal(ti.cpu().numpy(), ei)
                np.testing.assert_array_equal(tv.cpu().numpy(), ev)
            elif what == "cli":
                import pickle
                dense_only = bool(rng.random() < 0.25)
                lamda = float(rng.choice([1.0, 0.3, 2.5]))
                total = int(rng.choice([1, 1, 2, 3]))
                shrad = int(rng.integers(0, total))
                mode = str(rng.choice(["brute", "theta", "theta_rerank", "ip", "ip_rerank"]))
                lo_r, hi_r = O.shard_rows(n, total, shrad)
                n_sh = hi_r - lo_r
                topk = int(min(n_sh, rng.choice([1, 10, 100, 1000])))
                agip = int(min(n_sh, max(topk, rng.choice([10, 100, 1000, 5000]))))
                self_row = int(rng.integers(0, n))
                cfg.update(dense_only=dense_only, lamda=lamda, total_shrad=total, shrad=shrad, mode=mode, topk=topk, agip_topk=agip, self_row=self_row)
                if not live:
                    continue
                docids = ["d%d" % j for j in range(n)]
                qids = ["q%d" % j for j in range(q)]
                qids[0] = docids[self_row]                 # a query whose own document is in the corpus: its line is skipped, the rank numbers keep the gap
                qpath, cpath, opath = (os.path.join(tmp, f) for f in ("q.pt", "c.pt", "out.trec"))
                qv16 = qv.astype(np.float16)
                if dense_only:
                    if d_cls == 0:
                        continue
                    with open(qpath, "wb") as f:
                        pickle.dump([qv16[:, d_dlr:], None, qids], f, protocol=4)
                    with open(cpath, "wb") as f:
                        pickle.dump([cv[:, d_dlr:], 0, docids], f, protocol=4)     # a merged dense index stores the int 0 as its index array (index.py:40-43)
                else:
                    with open(qpath, "wb") as f:
                        pickle.dump([qv16, qi, qids], f, protocol=4)
                    with open(cpath, "wb") as f:
                        pickle.dump([cv, ci, docids], f, protocol=4)
                argv = ["--query_emb_path", qpath, "--index_path", cpath, "--emb_dim", str(0 if dense_only else d_dlr), "--topk", str(topk), "--agip_topk", str(agip),
                        "--lamda", str(lamda), "--total_shrad", str(total), "--shrad", str(shrad), "--output", opath, "--theta", str(theta)]
                argv += {"brute": ["--brute_force"], "theta": [], "theta_rerank": ["--rerank"], "ip": ["--IP"], "ip_rerank": ["--IP", "--rerank"]}[mode]
                G.main(argv)
                got = {}
                for line in open(opath).read().splitlines():
                    qid, q0, did, rank, sc_txt, run = line.split(" ")
                    assert q0 == "Q0" and run == "h2oloo"
                    got.setdefault(qid, []).append((int(did[1:]) - lo_r, int(rank), float(sc_txt)))
                qf, qif = O.prepare_queries(qv16[:, d_dlr:] if dense_only else qv16, None if dense_only else qi, 0 if dense_only else d_dlr, lamda)
                cs = cv[lo_r:hi_r, d_dlr:].astype(np.float32) if dense_only else c32[lo_r:hi_r]
                cis = None if dense_only else ci[lo_r:hi_r]
                for i, qid in enumerate(qids):
                    ex = O.gip_scores_f64(qf[i], None if dense_only else qif[i], cs, cis)
                    two_stage = (not dense_only) and mode != "brute"
                    s1 = O.stage1_scores_f64(qf[i], qif[i], cs, cis, theta, mode.startswith("ip")) if two_stage else ex
                    rerank = two_stage and mode.endswith("rerank")
                    k_out = min(topk, agip) if rerank else topk          # (no --rerank: the stage-1 list itself, topk long)
                    lst = got.get(qid, [])
                    rows_l = [-1] * k_out
                    sc_l = [0.0] * k_out
                    for row, rank, scv in lst:
                        assert 1 <= rank <= k_out and rows_l[rank - 1] == -1, ("rank", rank)
                        rows_l[rank - 1], sc_l[rank - 1] = row, scv
                    holes = [j for j in range(k_out) if rows_l[j] == -1]
                    if holes:                              # only the query's own document may be missing, at one rank
                        assert i == 0 and len(holes) == 1 and lo_r <= self_row < hi_r, ("holes", holes)
                        rows_l[holes[0]] = self_row - lo_r
                        sc_l[holes[0]] = float(np.float32((ex if (rerank or not two_stage) else s1)[self_row - lo_r]))
                    elif i == 0 and lo_r <= self_row < hi_r:
                        assert (self_row - lo_r) not in rows_l
                    assert all(0 <= r_ < n_sh for r_ in rows_l)
                    sc_a = np.asarray(sc_l, np.float32)
                    assert np.array_equal(sc_a.astype(np.float64), np.asarray(sc_l)), "a score in the file is not an fp32 value"
                    if rerank:
                        O.check_two_stage(rows_l, sc_a, s1, ex, agip, topk)
                    else:
                        O.check_topk(rows_l, sc_a, s1, k_out)
                for f_ in (qpath, cpath, opath):
                    os.unlink(f_)
            elif what == "strided":
                # the C structs carry leading dimensions: rows of wider host arrays / device tensors (column slices), fp16 and fp32 query batches,
                # against the same search on contiguous copies -- bit-equal
                padc, padq, padi = int(rng.choice([0, 8, 13])), int(rng.choice([0, 3, 64])), int(rng.choice([0, 1, 5]))
                kk = int(min(n, rng.choice([1, 10, 100, 1000])))
                on_dev = bool(rng.random() < 0.5)
                cfg.update(pad_corpus=padc, pad_query=padq, pad_index=padi, on_device=on_dev, k_search=kk)
                if not live:
                    continue
                def wide(a, pad, fill):
                    big = np.full((a.shape[0], a.shape[1] + pad), fill, a.dtype)
                    big[:, :a.shape[1]] = a
                    return big
                qsrc = qv if q32 else qv.astype(np.float16)
                bc, bci, bq, bqi = wide(cv, padc, 7), wide(ci, padi, 1), wide(qsrc, padq, 9), wide(qi, padi, 1)
                if on_dev:
                    bc, bci, bq, bqi = (torch.from_numpy(x).cuda() for x in (bc, bci, bq, bqi))
                vc, vci, vq, vqi = bc[:, :K], bci[:, :d_dlr], bq[:, :K], bqi[:, :d_dlr]
                ix = G.GipIndex(cv, ci, idx_buckets=nb)
                try:
                    s0, r0 = ix.search(qsrc, qi, kk)
                finally:
                    ix.close()
                ix = G.GipIndex(vc, vci, idx_buckets=nb)
                try:
                    s1, r1 = ix.search(vq, vqi, kk)
                    s2, r2 = ix.search_rerank(vq, vqi, vq, vqi, kk, kk)      # stage 1 = the full batch: the same lists again
                finally:
                    ix.close()
                for a_, b_ in ((r1, r0), (s1, s0), (r2, r0), (s2, s0)):
                    np.testing.assert_array_equal(np.asarray(a_), b_)
                for i in range(q):
                    O.check_topk(r0[i], s0[i], O.gip_scores_f64(qsrc[i].astype(np.float32), qi[i], c32, ci), kk)
            elif what == "params":
                # every tuning parameter of a handle at random values of its WHOLE admitted range (extremes included), on a corpus large enough for the
                # sampled controller: whatever the plan, the result is the exact top-k (values outside the range are refused: test_bad_arguments...)
                nn = int(rng.choice([20000, 70000, 150000]))
                dd, dc = int(rng.choice([32, 64, 128])), int(rng.choice([0, 32, 64]))
                kk = int(rng.choice([10, 100, 1000, 3000]))
                def pick(lo, hi, step=1):
                    r_ = rng.random()
                    return int(lo if r_ < 0.2 else hi if r_ < 0.4 else lo + step * rng.integers(0, (hi - lo) // step + 1))
                prm = {_lib.PARAM_CAND_CAP: pick(1024, 1 << 20), _lib.PARAM_FIRST_ROWS: pick(0, nn + 1000), _lib.PARAM_MAX_GROWTH: pick(1, 1024),
                       _lib.PARAM_SAMPLE_PERIOD: pick(0, 256), _lib.PARAM_MAIN_CHUNKS: pick(1, 64), _lib.PARAM_PROGRESSIVE_THR: pick(0, 2),
                       _lib.PARAM_AUX_CUS: pick(0, 192, 8), _lib.PARAM_GEMM_EXCLUSIVE: pick(0, 1), _lib.PARAM_OVERLAP_AUX: pick(-1, 1),
                       _lib.PARAM_SAMPLE_SHARE: pick(1, 4096), _lib.PARAM_ASYNC_CONTROLLER: pick(0, 2), _lib.PARAM_LIST_STRIDE: 0 if rng.random() < 0.4 else pick(1024, 1 << 20, 256),
                       _lib.PARAM_PROFILE: pick(0, 1)}
                keys = [k_ for k_ in prm if rng.random() < 0.6]
                cvp = np.abs(rng.standard_normal((nn, dd + dc), dtype=np.float32)) * 0.3
                cvp[:, dd:] = rng.standard_normal((nn, dc), dtype=np.float32) * 0.1
                cvp = cvp.astype(np.float16)
                qvp = np.abs(rng.standard_normal((q, dd + dc), dtype=np.float32)) * 0.3
                qvp[:, dd:] = rng.standard_normal((q, dc), dtype=np.float32) * 0.1
                qvp = qvp.astype(np.float16).astype(np.float32)
                cip = rng.integers(0, 7, (nn, dd)).astype(np.uint8)
                qip = rng.integers(0, 7, (q, dd)).astype(np.uint8)
                cfg.update(n=nn, d_dlr=dd, d_cls=dc, k_search=kk, params={int(k_): prm[k_] for k_ in keys})
                if not live:
                    continue
                if os.environ.get("DHR_STRESS_DUMP"):          # replay aid: the inputs of this case for a stand-alone script
                    np.savez(os.environ["DHR_STRESS_DUMP"], cv=cvp, ci=cip, qv=qvp, qi=qip, k=kk, nb=nb, g8=g8, keys=np.array(keys), vals=np.array([prm[k_] for k_ in keys]))
                ix = G.GipIndex(cvp, cip, idx_buckets=nb)
                try:
                    for k_ in keys:
                        ix.set_param(k_, prm[k_])
                    sp, rp = ix.search(qvp, qip, kk)
                    sp2, rp2 = ix.search(qvp, qip, kk)             # ... twice: the second call starts from the first one's workspace
                finally:
                    ix.close()
                np.testing.assert_array_equal(rp2, rp)
                np.testing.assert_array_equal(sp2, sp)
                cp32 = cvp.astype(np.float32)
                for i in range(q):
                    ex = O.gip_scores_f64(qvp[i], qip[i], cp32, cip)
                    O.check_topk(rp[i], sp[i], ex, kk)
                    np.testing.assert_allclose(sp[i], ex[rp[i]].astype(np.float32), rtol=0, atol=1e-6 * max(1.0, np.abs(ex).max()))
            elif what == "pq":
                from dhr_amd.retrieval import quantize_index as QI
                from oracle import pq_oracle as PO
                M = int(rng.choice([1, 4, 8, 16, 64]))
                dsub = int(rng.choice([1, 2, 7, 14, 24]))
                nbits = int(rng.choice([4, 6, 8]))
                kp = int(min(n, rng.choice([1, 10, 500, 5000, 20000])))
                cb = (rng.standard_normal((M, 1 << nbits, dsub)) * rng.choice([0.1, 1.0, 30.0])).astype(np.float32)
                codes = rng.integers(0, 1 << nbits, (n, M)).astype(np.uint8)
                if rng.random() < 0.3:
                    codes[:, :] = codes[0]                 # every row the same score: ties down to "row asc"
                qp = rng.standard_normal((q, M * dsub)).astype(np.float32)
                xs = rng.standard_normal((min(n, 400), M * dsub)).astype(np.float16).astype(np.float32)      # (fp32 in: the wrapper converts back to the records' fp16)
                cfg.update(M=M, dsub=dsub, nbits=nbits, k_pq=kp)
                if not live:
                    continue
                pix = QI.PqIndex(cb, codes, nbits=nbits)
                try:
                    adc = PO.adc_scores(qp, codes, cb)
                    tol = 1e-5 * max(1.0, float(np.abs(adc).max()))
                    np.testing.assert_allclose(pix.adc_scores(qp).cpu().numpy(), adc, rtol=0, atol=tol)
                    sp, rp = pix.search(qp, kp)
                    for i in range(q):
                        O.check_topk(rp[i], sp[i], adc[i], kp, atol=10 * tol)
                finally:
                    pix.close()
                np.testing.assert_array_equal(QI.decode(cb, codes[:400]), PO.decode(codes[:400], cb).astype(np.float16))
                enc, ref = QI.encode(xs, cb, nbits), PO.encode(xs, cb)
                if (enc != ref).any():                     # a different code is admissible only where its centroid is as near (float ties)
                    sub = xs.reshape(len(xs), M, dsub)
                    for a, m in zip(*np.nonzero(enc != ref)):
                        d0 = float(((sub[a, m] - cb[m, enc[a, m]]) ** 2).sum()); d1 = float(((sub[a, m] - cb[m, ref[a, m]]) ** 2).sum())
                        assert abs(d0 - d1) <= 1e-4 * max(1.0, d1), (a, m, d0, d1)
                path = os.path.join(tmp, "pq.idx")
                QI.save_pq(path, cb, codes, nbits)
                back = QI.load_pq(path)
                os.unlink(path)
                assert back["nbits"] == nbits
                np.testing.assert_array_equal(back["codes"], codes)
                np.testing.assert_array_equal(np.asarray(back["codebooks"], np.float32).reshape(cb.shape), cb)
            elif what == "local_shards":
                n_sh = int(rng.integers(1, 6))
                cuts = np.sort(rng.choice(np.arange(1, n), n_sh - 1, replace=False)) if n_sh > 1 else np.zeros(0, np.int64)
                bounds = [0] + [int(c) for c in cuts] + [n]
                if not live:
                    continue
                shards = [G.GipIndex(cv[a:b], ci[a:b], row_offset=a, idx_buckets=nb) for a, b in zip(bounds[:-1], bounds[1:])]
                try:
                    kk = k1
                    s, r = D.search_sharded_local(shards, qv, qi, kk)
                    s, r = np.asarray(s.cpu() if hasattr(s, "cpu") else s), np.asarray(r.cpu() if hasattr(r, "cpu") else r)
                    for i in range(q):
                        ex = exact(i)
                        O.check_topk(r[i], s[i], ex, kk)
                        np.testing.assert_allclose(s[i], ex[r[i]].astype(np.float32), rtol=0, atol=1e-6 * max(1.0, np.abs(ex).max()))
                finally:
                    for sh in shards:
                        sh.close()
                cfg["bounds"] = bounds
            else:
                if what == "score_rows":
                    m = int(rng.choice([1, 3, 100, 1000, 5000]))
                    rows = rng.integers(0, n, (q, m)).astype(np.int64)          # duplicates allowed
                if not live:
                    continue
                ix = G.GipIndex(cv, ci, idx_buckets=nb)
                try:
                    if what in ("theta", "ip"):
                        if what == "theta":
                            q1, qi1 = np.where(qv > theta, qv, np.float32(0)), qi
                        else:
                            q1, qi1 = qv, None
                        s, r = ix.search_rerank(q1, qi1, qv, qi, k1, k)
                        for i in range(q):
                            s1 = O.stage1_scores_f64(qv[i], qi[i], c32, ci, theta, what == "ip")
                            O.check_two_stage(r[i], s[i], s1, exact(i), k1, k)
                    elif what == "score_rows":
                        got = ix.score_rows(qv, qi, rows)
                        for i in range(q):
                            ex = exact(i)
                            np.testing.assert_allclose(got[i], ex[rows[i]].astype(np.float32), rtol=0, atol=1e-6 * max(1.0, np.abs(ex).max()))
                    else:
                        kk = k1
                        s0, r0 = ix.search(qv, qi, kk)
                        path = os.path.join(tmp, "ix.dhr")
                        ix.save(path, docids=["d%d" % j for j in range(min(n, 50))])
                        ix2, ids = G.GipIndex.load(path)
                        try:
                            s1, r1 = ix2.search(qv, qi, kk)
                        finally:
                            ix2.close()
                        os.unlink(path)
                        assert ids == ["d%d" % j for j in range(min(n, 50))]
                        np.testing.assert_array_equal(r1, r0)
                        np.testing.assert_array_equal(s1, s0)
                        for i in range(q):
                            O.check_topk(r0[i], s0[i], exact(i), kk)
                finally:
                    ix.close()
        except Exception as e:  # noqa: BLE001
            print("FAILED", cfg, "->", repr(e)[:600])
            sys.exit(1)
        if live and (case % 10 == 0 or only >= 0):
            print("case %d ok (%.0f s) %s" % (case, time.time() - t0, cfg), flush=True)
    print("all %d cases ok in %.0f s" % (n_cases, time.time() - t0))


if __name__ == "__main__":
    main()
