#!/usr/bin/env python
"""State-machine fuzz of the staged search (the calls dist.py and dhr_search_sharded are built from: dhr_search_pre / _begin / _begin_rest / _mid / _finish)
mixed with the calls that share the handle's workspace (dhr_search, dhr_search_rerank, dhr_score_rows, dhr_index_set_param): random call sequences in
ANY order with plausible and implausible thresholds.  Every call either succeeds or returns a status (DhrError) -- no crash, no hang -- and whatever came
before, (a) a staged search that follows the protocol with admissible thresholds returns the exact top-k, (b) a plain search afterwards equals the reference.
usage: python tools/fuzz_staged.py [n_sequences] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    from dhr_amd import _lib, synth
    from dhr_amd.retrieval import gip_retrieval as G
    from oracle import gip_oracle as O
    rng = np.random.default_rng(seed)
    t0 = time.time()
    stats = dict(ok=0, refused=0, protocol=0)
    for rep in range(max(1, n_seq // 25)):
        n = int(rng.choice([3000, 40000, 120000]))
        nq, k = int(rng.integers(1, 10)), int(rng.choice([10, 100, 1000]))
        d_dlr, d_cls = int(rng.choice([0, 64, 128])), int(rng.choice([32, 64]))
        cv, ci, qv, qi = synth.make_pair(int(rng.integers(1, 1 << 30)), n, nq, d_dlr, d_cls) if d_dlr else synth.make_pair(int(rng.integers(1, 1 << 30)), n, nq, 0, d_cls, kind="dense")
        q32 = qv.astype(np.float32)
        c32 = cv.astype(np.float32)
        exact = np.stack([O.gip_scores_f64(q32[i], None if ci is None else qi[i], c32, ci) for i in range(nq)])
        kth = np.sort(exact, axis=1)[:, ::-1][:, min(k, n) - 1]
        ix = G.GipIndex(cv, ci)
        share = [int(rng.choice([1, 2, 8]))]
        ix.set_param(_lib.PARAM_SAMPLE_SHARE, share[0])
        s_ref, r_ref = ix.search(q32, qi, min(k, n))
        dev = torch.device("cuda", 0)

        def tau(kind):
            if kind == 0:
                return torch.full((nq,), float("-inf"), device=dev)
            if kind == 1:
                return torch.from_numpy((kth - 1e-3 * np.abs(kth) - 1e-6).astype(np.float32)).to(dev)       # admissible: just below the true k-th score
            if kind == 2:
                return torch.full((nq,), 1e30, device=dev)                                                # nothing passes
            return torch.from_numpy(rng.standard_normal(nq).astype(np.float32)).to(dev)
        for _ in range(25):
            seq = [str(rng.choice(["begin", "pre", "begin_rest", "mid", "finish", "search", "score_rows", "rerank", "share", "ranks"])) for _ in range(int(rng.integers(1, 9)))]
            if rng.random() < 0.35:                        # a protocol-conforming staged search with admissible thresholds: must be exact
                seq = [str(rng.choice(["begin", "pre+rest"])), str(rng.choice(["mid", "none"])), "finish_ok"]
            loose = bool(rng.random() < 0.3)               # protocol sequences: thresholds of -inf instead of admissible ones (lists may overflow: flagged, not wrong)
            log = []
            try:
                for op in seq:
                    try:
                        if op == "begin":
                            ix.search_begin(q32, qi, min(k, n))
                        elif op == "pre":
                            ix.search_pre(q32, qi, min(k, n), r_local=int(rng.integers(0, 40)))
                        elif op == "pre+rest":
                            if ix.pre_ranks(min(k, n))[0] > 0:
                                ix.search_pre(q32, qi, min(k, n))
                                ix.search_begin_rest(tau(0 if loose else 1))
                            else:
                                ix.search_begin(q32, qi, min(k, n))
                        elif op == "begin_rest":
                            ix.search_begin_rest(tau(int(rng.integers(0, 4))))
                        elif op == "mid":
                            kind = (0 if loose else 1) if seq[-1] == "finish_ok" else int(rng.integers(0, 4))
                            if seq[-1] != "finish_ok" or ix.mid_ranks(min(k, n))[0] > 0:
                                ix.search_mid(tau(kind), r_local=0 if seq[-1] == "finish_ok" else int(rng.integers(0, 70)))
                        elif op == "finish":
                            ix.search_finish(None if rng.random() < 0.2 else tau(int(rng.integers(0, 4))))
                        elif op == "finish_ok":
                            s, r, cnt = ix.search_finish(tau(0 if loose else 1))
                            s, r, cnt = s.cpu().numpy(), r.cpu().numpy(), cnt.cpu().numpy()
                            kk = min(k, n)
                            # count = rows of this shard reaching the threshold (-1: a list overflowed -- the caller repairs that query): with thresholds just
                            # below the true k-th score every query reports >= k rows and the exact top-k; a flagged query is admissible only under -inf, or where
                            # the handle was told that it is one of several shards (DHR_PARAM_SAMPLE_SHARE > 1: its lists are sized for its share of k, and this
                            # one holds all of the top-k -- the skewed-shard case the caller's repair step exists for)
                            assert loose or share[0] > 1 or (cnt >= kk).all(), cnt
                            for i in range(nq):
                                if cnt[i] >= kk:
                                    O.check_topk(r[i], s[i], exact[i], kk)
                                else:
                                    assert cnt[i] == -1, cnt
                            stats["protocol"] += 1
                        elif op == "search":
                            s, r = ix.search(q32, qi, min(k, n))
                            np.testing.assert_array_equal(r, r_ref); np.testing.assert_array_equal(s, s_ref)
                        elif op == "score_rows":
                            got = ix.score_rows(q32, qi, r_ref[:, :5])
                            np.testing.assert_array_equal(got, s_ref[:, :5])
                        elif op == "rerank":
                            ix.search_rerank(q32, qi, q32, qi, min(n, 2 * k), min(k, n))
                        elif op == "share":
                            share[0] = int(rng.choice([1, 2, 8, 64]))
                            ix.set_param(_lib.PARAM_SAMPLE_SHARE, share[0])
                        elif op == "ranks":
                            ix.sample_rank(min(k, n)); ix.union_rank(min(k, n)); ix.pre_ranks(min(k, n)); ix.mid_ranks(min(k, n))
                        elif op == "none":
                            pass
                        log.append(op)
                        stats["ok"] += 1
                    except _lib.DhrError as e:
                        log.append(op + "!" + str(e.status))
                        stats["refused"] += 1
                        if seq[-1] == "finish_ok":
                            raise
                torch.cuda.synchronize()
                s, r = ix.search(q32, qi, min(k, n))
                np.testing.assert_array_equal(r, r_ref); np.testing.assert_array_equal(s, s_ref)
            except Exception as e:  # noqa: BLE001
                print("FAILED", dict(n=n, nq=nq, k=k, d_dlr=d_dlr, d_cls=d_cls, seq=seq, log=log, loose=loose, share=share[0]), "->", repr(e)[:600])
                sys.exit(1)
        ix.close()
        print("corpus %d ok (%.0f s): n %d q %d k %d dims %d+%d  %s" % (rep, time.time() - t0, n, nq, k, d_dlr, d_cls, stats), flush=True)
    print("all sequences ok in %.0f s: %s" % (time.time() - t0, stats))


if __name__ == "__main__":
    main()
