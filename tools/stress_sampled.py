#!/usr/bin/env python
"""Randomised differential test of the CONTROLLER: corpora large enough for the sampled thresholds (70 k ... 500 k rows),
random sample period / list capacity / chunk count / head size, benign and adversarial row orders (sorted by score,
all good rows in sample tiles, all in one non-sample residue), single index and the staged (sharded) search with the
common threshold -- against the oracle's float64 scores.  Prints the failing configuration and exits non-zero on the
first mismatch.  usage: python tools/stress_sampled.py [n_cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    from dhr_amd import _lib, dist as D
    from dhr_amd.retrieval import gip_retrieval as G
    from oracle import gip_oracle as O
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n_sampled = n_fallback = 0
    for case in range(n_cases):
        n = int(rng.choice([70_000, 150_000, 300_000, 500_000]))
        q = int(rng.integers(1, 24))
        d_dlr = int(rng.choice([0, 32, 64, 128]))
        d_cls = int(rng.choice([0, 32, 64])) if d_dlr else int(rng.choice([32, 64, 128]))
        k = int(rng.choice([10, 100, 1000]))
        period = int(rng.choice([0, 2, 4, 8, 16, 32]))
        cap = int(rng.choice([0, 0, 1024, 4096]))
        chunks = int(rng.choice([0, 1, 2, 5]))
        first = int(rng.choice([0, 0, 256, 2048]))
        order = str(rng.choice(["random", "random", "sorted_desc", "sorted_asc", "sample_tiles", "one_residue"]))
        shards = int(rng.choice([1, 1, 2, 3]))
        nb = int(rng.choice([0, 0, 1, 2]))
        g8 = int(rng.integers(0, 2))                   # image of the gated half: fp16 2:4 (the library's choice at these sizes) or int8 2:4
        os.environ["DHR_GATED_I8"] = str(g8)
        cfg = dict(case=case, n=n, q=q, d_dlr=d_dlr, d_cls=d_cls, k=k, period=period, cap=cap, chunks=chunks, first=first,
                   order=order, shards=shards, nb=nb, gated_i8=g8)
        K = d_dlr + d_cls
        cv = np.abs(rng.standard_normal((n, K), dtype=np.float32)) * 0.3
        cv[:, d_dlr:] = rng.standard_normal((n, d_cls), dtype=np.float32) * 0.1
        qv = np.abs(rng.standard_normal((q, K), dtype=np.float32)) * 0.3
        qv[:, d_dlr:] = rng.standard_normal((q, d_cls), dtype=np.float32) * 0.1
        cv = cv.astype(np.float16)
        qv = qv.astype(np.float16).astype(np.float32)
        ci = rng.integers(0, 7, (n, d_dlr)).astype(np.uint8) if d_dlr else None
        qi = rng.integers(0, 7, (q, d_dlr)).astype(np.uint8) if d_dlr else None
        if order != "random":
            # a per-row boost on a column every query weighs positively makes the row order matter for every query
            col = d_dlr                                    # first dense column (every config has d_cls > 0 or d_dlr > 0)
            if d_cls == 0:
                col = 0
                ci[:, 0] = 1; qi[:, 0] = 1
            qv[:, col] = 1.0
            boost = np.zeros(n, np.float32)
            tile = np.arange(n) // 256
            if order == "sorted_desc":
                boost = np.linspace(2.0, 0.0, n, dtype=np.float32)
            elif order == "sorted_asc":
                boost = np.linspace(0.0, 2.0, n, dtype=np.float32)
            else:
                p = max(period, 2)
                head_tiles = 0        # no exhaustive head in a sampled search since round 5 (threshold bootstrap)
                res = 0 if order == "sample_tiles" else min(3, p - 1)
                boost[(tile >= head_tiles) & (((tile - head_tiles) % p) == res)] = 2.0
            cv[:, col] = (cv[:, col].astype(np.float32) * 0.1 + boost).astype(np.float16)
        try:
            c32 = cv.astype(np.float32)
            def make(lo, hi):
                ix = G.GipIndex(cv[lo:hi], None if ci is None else ci[lo:hi], row_offset=lo, idx_buckets=nb if d_dlr else 0)
                ix.set_param(_lib.PARAM_SAMPLE_PERIOD, period)
                if cap:
                    ix.set_param(_lib.PARAM_CAND_CAP, cap)
                if chunks:
                    ix.set_param(_lib.PARAM_MAIN_CHUNKS, chunks)
                if first:
                    ix.set_param(_lib.PARAM_FIRST_ROWS, first)
                return ix
            if shards == 1:
                ix = make(0, n)
                s, r = ix.search(qv, qi, k)
                st = ix.stats()
                n_sampled += int(ix.sample_rank(k) > 0)
                n_fallback += int(st["sample_fallback_queries"] > 0)
                ix.close()
            else:
                parts = [make(*D.shard_bounds(n, shards, sh)) for sh in range(shards)]
                rr = [p.sample_rank(k) for p in parts]
                if min(rr) == max(rr) and rr[0] > 0:
                    n_sampled += 1
                    samples = [p.search_begin(qv, qi, k) for p in parts]
                    tau = D.common_threshold(torch.stack(samples), rr[0])
                    outs = [p.search_finish(tau) for p in parts]
                    count = torch.stack([o[2] for o in outs])
                    failed = torch.nonzero((count.clamp(min=0).sum(0) < k) | (count < 0).any(0)).flatten()
                    scores = [o[0] for o in outs]; rows = [o[1] for o in outs]
                    if failed.numel():
                        n_fallback += 1
                        ids = failed.cpu().numpy()
                        for i, p in enumerate(parts):
                            fs, fr = p.search(qv[ids], None if qi is None else qi[ids], k, out_device=True)
                            scores[i][failed] = fs; rows[i][failed] = fr
                    ms, mr = D.merge_sorted_lists(torch.stack(scores), torch.stack(rows), k)
                else:
                    outs = [p.search(qv, qi, k, out_device=True) for p in parts]
                    ms, mr = D.merge_sorted_lists(torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), k)
                s, r = ms.cpu().numpy(), mr.cpu().numpy()
                for p in parts:
                    p.close()
            for i in range(q):
                ex = O.gip_scores_f64(qv[i], None if qi is None else qi[i], c32, ci)
                O.check_topk(r[i], s[i], ex, k)
                np.testing.assert_allclose(s[i], ex[r[i]].astype(np.float32), rtol=0, atol=1e-6 * max(1.0, np.abs(ex).max()))
        except Exception as e:  # noqa: BLE001
            print("FAILED", cfg, "->", repr(e)[:600])
            sys.exit(1)
        if case % 5 == 0:
            print("case %d ok (%.0f s) %s" % (case, time.time() - t0, cfg), flush=True)
    print("all %d cases ok in %.0f s (sampled path in %d, fallback taken in %d)" % (n_cases, time.time() - t0, n_sampled, n_fallback))


if __name__ == "__main__":
    main()
