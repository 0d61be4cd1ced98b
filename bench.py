#!/usr/bin/env python
"""Benchmark of the brute-force dense-hybrid retrieval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: ALL queries of the batch (6 980, MS MARCO
dev.small) searched against the whole corpus (8 841 823 synthetic passages, 768 densified-lexical +
768 dense columns = BASELINE.json config 3, "DeLADE-CLS 768+768 dense-hybrid, 1xMI355X"), exact
top-1000.  With N>1 the corpus is row-sharded over the ranks (gip_retrieval.py:292-306 arithmetic),
every rank searches its shard and ONE RCCL all-gather + k-way reduce produces the global top-k on
every rank (total work fixed -> "strong" scaling).  Corpus and queries are resident in HBM before
the timed region (the reference also times only its query loop, gip_retrieval.py:107,161-163).

Prints ONE JSON line (rank 0).  See DESIGN.md section "Measurement" for the field definitions.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MSMARCO = 8_841_823
Q_DEV = 6_980
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md chip table


def gen_shard(torch, synth, device, seed, n_rows, d_dlr, d_cls, lmin, lmax, uniform_idx, chunk=1 << 18):
    """Synthetic rows on the GPU, SURVEY.md section 8(d) recipe.  -> (value fp16 [n,K], index u8 [n,d_dlr]|None)"""
    gen = torch.Generator(device=device).manual_seed(seed)
    k = d_dlr + d_cls
    value = torch.empty((n_rows, k), dtype=torch.float16, device=device)
    index = torch.empty((n_rows, d_dlr), dtype=torch.uint8, device=device) if d_dlr else None
    for lo in range(0, n_rows, chunk):
        hi = min(n_rows, lo + chunk)
        if d_dlr:
            v, i = synth.torch_make_dlr(gen, hi - lo, d_dlr, lmin, lmax, device, uniform_idx=uniform_idx)
            value[lo:hi, :d_dlr] = v
            index[lo:hi] = i
        if d_cls:
            value[lo:hi, d_dlr:] = (torch.randn((hi - lo, d_cls), generator=gen, device=device) * 0.1).to(torch.float16)
    return value, index


def cpu_baseline(O, cv, ci, qv, qi, k, n_full):
    """The oracle's restatement of the reference's per-query loop (mask*corpus -> einsum -> topk,
    gip_retrieval.py:119-125), fp32, ONE thread (the reference sets torch.set_num_threads(1) for
    --batch 1, :255-259), on a bounded sample; linear extrapolation in the row count."""
    c32 = cv.astype(np.float32)
    q32 = qv.astype(np.float32)
    if ci is not None:
        cls_dim = c32.shape[1] - ci.shape[1]
        cip = O.pad_idx(ci, cls_dim)
        qip = O.pad_idx(qi, cls_dim)
    t0 = time.perf_counter()
    rows = []
    for i in range(q32.shape[0]):
        s = O.gip_scores_f32(q32[i], qip[i], c32, cip) if ci is not None else O.ip_scores_f32(q32[i], c32)
        rows.append(O.topk_desc(s, k))
    dt = time.perf_counter() - t0
    s_per_query = dt / q32.shape[0]
    return s_per_query, s_per_query * (n_full / c32.shape[0]), rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="hybrid", choices=["hybrid", "dense"],
                    help="hybrid = config 3 (768 DLR + 768 dense); dense = config 2 (768 dense, no index array)")
    ap.add_argument("--n-docs", type=int, default=N_MSMARCO)
    ap.add_argument("--n-queries", type=int, default=Q_DEV)
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--uniform-idx", action="store_true", help="adversarial variant: uniform slice indices")
    ap.add_argument("--cand-cap", type=int, default=0)
    ap.add_argument("--idx-buckets", type=int, default=0)
    ap.add_argument("--sample-period", type=int, default=-1)
    ap.add_argument("--main-chunks", type=int, default=0)
    ap.add_argument("--max-growth", type=int, default=0, help="tuning: max (next sampled chunk) / (rows seen) in 1/16ths")
    ap.add_argument("--aux-cus", type=int, default=-1, help="tuning: CUs the refine/rescoring stream is confined to (0 = no mask)")
    ap.add_argument("--gemm-exclusive", type=int, default=-1)
    ap.add_argument("--overlap-aux", type=int, default=-1)
    ap.add_argument("--gemm-variant", type=int, default=-1, help="tuning: bound-GEMM kernel of the 2:4 layout (3 = 12-wave producer / consumer, 4 = 4-wave)")
    ap.add_argument("--no-progressive-thr", action="store_true", help="A/B: keep the sampled thresholds frozen over the main pass")
    ap.add_argument("--first-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=200_000)
    ap.add_argument("--cpu-queries", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1237)
    ap.add_argument("--dist-backend", default="nccl", help="testing only: 'gloo' lets several ranks share ONE GPU (with DHR_BENCH_SINGLE_DEVICE=1)")
    args = ap.parse_args()

    import torch
    from dhr_amd import _lib, dist as D, synth
    from dhr_amd.retrieval.gip_retrieval import GipIndex

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if os.environ.get("DHR_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0                                   # testing only: every rank on device 0 (needs --dist-backend gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)

    d_dlr, d_cls = (768, 768) if args.workload == "hybrid" else (0, 768)
    K = d_dlr + d_cls
    n, nq, k = args.n_docs, args.n_queries, args.topk
    lo, hi = D.shard_bounds(n, world, rank)

    # ---------------- data (outside the timed region)
    t_gen = time.perf_counter()
    cv, ci = gen_shard(torch, synth, device, args.seed + 1000 * rank, hi - lo, d_dlr, d_cls, 30, 90, args.uniform_idx)
    qv, qi = gen_shard(torch, synth, device, args.seed + 999_983, nq, d_dlr, d_cls, 4, 12, args.uniform_idx)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    t_build = time.perf_counter()
    index = GipIndex(cv, ci, device=local_rank, row_offset=lo, idx_buckets=args.idx_buckets)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    index.set_param(_lib.PARAM_PROFILE, 1)
    if args.cand_cap:
        index.set_param(_lib.PARAM_CAND_CAP, args.cand_cap)
    if args.sample_period >= 0:
        index.set_param(_lib.PARAM_SAMPLE_PERIOD, args.sample_period)
    if args.max_growth:
        index.set_param(_lib.PARAM_MAX_GROWTH, args.max_growth)
    if args.aux_cus >= 0:
        index.set_param(_lib.PARAM_AUX_CUS, args.aux_cus)
    if args.overlap_aux >= 0:
        index.set_param(_lib.PARAM_OVERLAP_AUX, args.overlap_aux)
    if args.gemm_variant >= 0:
        index.set_param(_lib.PARAM_GEMM_VARIANT, args.gemm_variant)
    if args.gemm_exclusive >= 0:
        index.set_param(_lib.PARAM_GEMM_EXCLUSIVE, args.gemm_exclusive)
    if args.no_progressive_thr:
        index.set_param(_lib.PARAM_PROGRESSIVE_THR, 0)
    if args.main_chunks:
        index.set_param(_lib.PARAM_MAIN_CHUNKS, args.main_chunks)
    if args.first_rows:
        index.set_param(_lib.PARAM_FIRST_ROWS, args.first_rows)

    # host sample for the CPU baseline + an in-bench parity check (rank 0, N=1 only)
    sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        m, mq = min(args.cpu_rows, hi - lo), min(args.cpu_queries, nq)
        sample = (cv[:m].cpu().numpy(), None if ci is None else ci[:m].cpu().numpy(),
                  qv[:mq].cpu().numpy(), None if qi is None else qi[:mq].cpu().numpy())
    del cv, ci
    torch.cuda.empty_cache()

    def step():
        if world > 1:
            return D.sharded_search(index, qv, qi, k)        # common thresholds + one all-gather of the shard lists
        return index.search(qv, qi, k, out_device=True)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gemm_ms = gemm_flops_alg = 0.0
    launches = 0
    stats_acc = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = index.stats()
        gemm_ms += st["gemm_ms"]
        launches += st["phases"]
        gemm_flops_alg += st["gemm_flops_alg"]                  # algorithmic: real Q and K of every launch
        for key in ("refine_ms", "rescore_ms", "select_ms", "prep_ms", "total_ms", "candidates_bound", "candidates_exact",
                    "overflow_retries", "gemm_rows", "sample_fallback_queries"):
            stats_acc[key] = stats_acc.get(key, 0) + st[key]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    qps = nq * args.steps / elapsed

    # ---- N > 1: size-independent checks of the sharded result, outside the timed region (the oracle cannot hold the
    # corpus): sorted lists of distinct rows; every returned score is the exact score of its row on the rank that holds
    # the row (dhr_score_rows, an independent code path); no sampled row of any shard outside a list beats its k-th score
    dist_check = None
    if world > 1:
        import torch.distributed as dist
        gs, gr = step()
        torch.cuda.synchronize()
        ds = gs[:, 1:] - gs[:, :-1]
        ok_sorted = bool((ds <= 0).all()) and bool((gr[:, 1:][ds == 0] > gr[:, :-1][ds == 0]).all())
        ok_distinct = int(torch.sort(gr, dim=1).values.diff(dim=1).eq(0).sum()) == 0
        sub = torch.arange(0, nq, max(1, nq // 32), device=device)
        qs = qv[sub].cpu().numpy().astype(np.float32)
        qis = None if qi is None else qi[sub].cpu().numpy()
        rows = gr[sub].cpu().numpy()
        mine = torch.from_numpy(index.score_rows(qs, qis, rows)).to(device)     # -inf for the rows of other shards
        dist.all_reduce(mine, op=dist.ReduceOp.MAX)
        ok_scores = bool(torch.equal(mine, gs[sub]))
        g = torch.Generator(device="cpu").manual_seed(7 + rank)
        rnd = (torch.randint(0, hi - lo, (len(sub), 50_000), generator=g) + lo).numpy().astype(np.int64)
        rs = index.score_rows(qs, qis, rnd)
        kth = gs[sub, k - 1].cpu().numpy()[:, None]
        inlist = np.stack([np.isin(rnd[i], rows[i]) for i in range(len(sub))])
        beat = torch.tensor([int(np.count_nonzero((rs > kth) & ~inlist))], dtype=torch.int64, device=device)
        dist.all_reduce(beat, op=dist.ReduceOp.SUM)
        dist_check = {"queries": int(len(sub)), "sorted": ok_sorted, "distinct_rows": ok_distinct,
                      "scores_equal_exact_rescoring": ok_scores, "sampled_rows_per_rank": 50_000,
                      "rows_beating_kth_outside_list": int(beat.item())}
        if not (ok_sorted and ok_distinct and ok_scores and int(beat.item()) == 0):
            raise SystemExit("sharded result failed its property check: %s" % dist_check)

    out = None
    if rank == 0:
        ach_tf = gemm_flops_alg / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
        out = {
            "metric": "queries/sec (exact top-%d, brute-force dense-hybrid GIP retrieval)" % k,
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16 (fp16 x fp16 -> fp32 on the matrix cores for the bound; fp64-accumulated exact rescoring)", "data": "synthetic",
            "config": {"workload": ("MS MARCO-sized corpus %d x (%d DLR + %d dense) fp16%s, %d queries, top-%d, %s"
                                    % (n, d_dlr, d_cls, " + uint8 slice index" if d_dlr else "", nq, k,
                                       "uniform slice index (adversarial)" if args.uniform_idx else "densify-rule slice index")),
                       "baseline_config": "config 3: DeLADE-CLS 768+768 dense-hybrid" if d_dlr else "config 2: Aggretriever 768-d dense-only",
                       "parallelism": "rowshard%d+allgather" % world if world > 1 else "1gpu"},
            "roofline": {"bound": "mfma",
                         "kernel": ("gemm_filter_sparse_kernel (bound GEMM on the 2:4 sparse matrix cores + fused threshold filter)"
                                    if d_dlr and args.idx_buckets in (0, 2) else
                                    "gemm_filter_v3_kernel (bound GEMM + fused threshold filter)"),
                         "achieved": round(ach_tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach_tf / MFMA_PEAK_TFLOPS, 4),
                         # fabric-side (Infinity Cache + HBM) read bytes per launch: PMC passes cannot run inside this process,
                         # so this is the per-corpus-row figure of profiles/r01h_gemm_pmc.txt (FETCH_SIZE x2, rocprofv3 --pmc on
                         # the same kernel, 6 980 queries) x the average rows per launch; only quoted for that configuration
                         "traffic": (round(35.3e3 * stats_acc.get("gemm_rows", 0) / max(launches, 1), 0)
                                     if d_dlr == 768 and d_cls == 768 and nq == 6980 and args.idx_buckets in (0, 2) and world == 1 else None),
                         "traffic_unit": "bytes per launch (FETCH_SIZE, gfx950-corrected; profiles/r01h_gemm_pmc.txt)",
                         "launches": launches, "avg_launch_ms": round(gemm_ms / max(launches, 1), 3),
                         "alg_flops_per_step": gemm_flops_alg / args.steps},
            "whole_job_frac_of_gemm_roofline": round(qps * 2.0 * n * K / (MFMA_PEAK_TFLOPS * 1e12 * world), 4),
            "phase_ms_per_step": {key: round(v / args.steps, 3) for key, v in stats_acc.items() if key.endswith("_ms")}
                                 | {"gemm_ms": round(gemm_ms / args.steps, 3)},
            "candidates_per_query": {"bound": round(stats_acc["candidates_bound"] / args.steps / nq, 1),
                                     "exact": round(stats_acc["candidates_exact"] / args.steps / nq, 1)},
            "sample_fallback_queries_per_step": stats_acc["sample_fallback_queries"] / args.steps,
            "setup_s": {"generate": round(t_gen, 2), "index_build": round(t_build, 2)},
            "index_device_gb": round(index.device_bytes() / 1e9, 2),
        }
        if sample is not None:
            from oracle import gip_oracle as O
            scv, sci, sqv, sqi = sample
            s_q, s_q_full, cpu_rows = cpu_baseline(O, scv, sci, sqv, sqi, k, n)
            out["cpu_baseline"] = {"value": round(1.0 / s_q_full, 5), "unit": "queries/s", "cores": 1, "kind": "port",
                                   "sample": "%d queries x %d-row slice of the same synthetic corpus, fp32 numpy restatement of "
                                             "gip_retrieval.py:119-125, 1 thread; %.3f s/query measured, x%.1f linear extrapolation "
                                             "to %d rows" % (sqv.shape[0], scv.shape[0], s_q, n / scv.shape[0], n),
                                   "host_cpus": os.cpu_count()}
            # parity of the HIP path on the same sample (outside the timed region)
            six = GipIndex(scv, sci, device=local_rank)
            gs, gr = six.search(sqv.astype(np.float32), sqi, k)
            six.close()
            bad = 0
            for i in range(sqv.shape[0]):
                ex = O.gip_scores_f64(sqv[i].astype(np.float32), None if sqi is None else sqi[i], scv.astype(np.float32), sci)
                try:
                    O.check_topk(gr[i], gs[i], ex, k)
                except AssertionError:
                    bad += 1
            out["parity_check"] = {"queries": int(sqv.shape[0]), "rows": int(scv.shape[0]), "failed": bad}
        if dist_check is not None:
            out["parity_check"] = dist_check
        print(json.dumps(out), flush=True)
    index.close()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
