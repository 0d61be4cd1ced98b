#!/usr/bin/env python
"""Benchmark of the brute-force dense-hybrid retrieval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: ALL queries of the batch (6 980, MS MARCO
dev.small) searched against the whole corpus (8 841 823 synthetic passages, 768 densified-lexical +
768 dense columns = BASELINE.json config 3, "DeLADE-CLS 768+768 dense-hybrid, 1xMI355X"), exact
top-1000.  With N>1 the corpus is row-sharded over the ranks (gip_retrieval.py:292-306 arithmetic),
every rank searches its shard inside ONE library call (dhr_search_sharded) whose five small exchanges
run over RCCL: the ranks agree on the sample ranks (32 bytes), all-gather their sample scores
[Q, r] -> common thresholds, all-gather their best scores after the first eighth of the main pass
[Q, r2] -> raised thresholds, all-gather the per-query counts [Q], and all-gather the list prefixes
[Q, kk] (scores + rows) for the rank merge that leaves the global top-k on every rank (total work
fixed -> "strong" scaling).  The communicator is brought up under a watchdog (dhr_amd.dist.bring_up):
if RCCL fails or hangs on ANY rank, every rank degrades to torch.distributed gloo gathers on host
buffers -- the same control flow -- and the JSON line says so (`n_ranks_seen_by_rccl` = ncclCommCount
of the communicator the timed steps used, 0 for the host transport).  Corpus and queries are resident
in HBM before the timed region (the reference also times only its query loop,
gip_retrieval.py:107,161-163).

The default invocation (N=1, config 3) also times 5 steps each of config 2 (dense-only) and config 1
(BM25, 100 k rows) and attaches them as `other_configs` to the same JSON line, and 3 steps each of the
two-stage modes of the reference's docs on the resident config-3 index (`two_stage`).

Prints ONE JSON line (rank 0).  See DESIGN.md section "Measurement" for the field definitions.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_MSMARCO = 8_841_823
Q_DEV = 6_980
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md chip table
MFMA_PEAK_I8_TOPS = 5000.0     # dense int8 MFMA peak (same table): the ungated columns of a dense_i8 index run on it
SMFMAC_I8_PEAK_TOPS = 6650.0   # v_smfmac_i32_32x32x64_i8, logical (2 bucket columns per slice): 48 cycles per issue measured with zero operands (profiles/r03_smfmac8_probe.txt)
GEN_CHUNK = 1 << 18            # the synthetic corpus is seeded per GLOBAL chunk of this many rows: a shard of any world size is a slice of the same corpus

# BASELINE.json config 5: the 13 public BEIR corpora (documents, test queries), sizes from the public BEIR table
BEIR = [("trec-covid", 171_332, 50), ("nfcorpus", 3_633, 323), ("nq", 2_681_468, 3_452), ("hotpotqa", 5_233_329, 7_405),
        ("fiqa", 57_638, 648), ("arguana", 8_674, 1_406), ("webis-touche2020", 382_545, 49), ("quora", 522_931, 10_000),
        ("dbpedia-entity", 4_635_922, 400), ("scidocs", 25_657, 1_000), ("fever", 5_416_568, 6_666),
        ("climate-fever", 5_416_593, 1_535), ("scifact", 5_183, 300)]


def gen_rows(torch, synth, device, seed, row_lo, row_hi, d_dlr, d_cls, lmin, lmax, uniform_idx, clustered=None, hot_frac=0.0, dup=False):
    """Rows [row_lo, row_hi) of the synthetic matrix with this seed, on the GPU (SURVEY.md section 8(d) recipe).  Every global
    GEN_CHUNK-row chunk has its own generator seed, so the rows do not depend on how the corpus is sharded.
    -> (value fp16 [n,K], index u8 [n,d_dlr]|None)"""
    n_rows = row_hi - row_lo
    k = d_dlr + d_cls
    value = torch.empty((n_rows, k), dtype=torch.float16, device=device)
    index = torch.empty((n_rows, d_dlr), dtype=torch.uint8, device=device) if d_dlr else None
    for c in range(row_lo // GEN_CHUNK, (row_hi + GEN_CHUNK - 1) // GEN_CHUNK):
        c_lo = c * GEN_CHUNK
        a, b = max(row_lo, c_lo), min(row_hi, c_lo + GEN_CHUNK)
        gen = torch.Generator(device=device).manual_seed(seed * 1_000_003 + c)
        if d_dlr:
            v, i = synth.torch_make_dlr(gen, GEN_CHUNK, d_dlr, lmin, lmax, device, uniform_idx=uniform_idx)
            value[a - row_lo:b - row_lo, :d_dlr] = v[a - c_lo:b - c_lo]
            index[a - row_lo:b - row_lo] = i[a - c_lo:b - c_lo]
            del v, i
        if d_cls:
            if clustered is not None:      # --data clustered: clustered / anisotropic dense columns (dhr_amd/synth.py), the same chunk seeding
                d = synth.torch_make_dense_clustered(gen, GEN_CHUNK, d_cls, clustered, device, hot_frac=hot_frac)
            else:
                d = (torch.randn((GEN_CHUNK, d_cls), generator=gen, device=device) * 0.1).to(torch.float16)
            value[a - row_lo:b - row_lo, d_dlr:] = d[a - c_lo:b - c_lo]
            del d
        if dup and clustered is not None and c_lo >= row_lo and c_lo + GEN_CHUNK <= row_hi:
            # near-duplicate rows inside whole chunks of this shard (a chunk cut by a shard boundary is left alone: the duplicates of a
            # chunk are drawn over the whole chunk, and a shard must stay a slice of the same corpus)
            synth.torch_near_duplicates(gen, value[c_lo - row_lo:c_lo - row_lo + GEN_CHUNK], None if index is None else index[c_lo - row_lo:c_lo - row_lo + GEN_CHUNK], d_dlr)
    return value, index


def gen_shard(torch, synth, device, seed, n_rows, d_dlr, d_cls, lmin, lmax, uniform_idx, chunk=GEN_CHUNK):
    """Rows [0, n_rows) (kept for the tools that build one stand-alone shard)."""
    return gen_rows(torch, synth, device, seed, 0, n_rows, d_dlr, d_cls, lmin, lmax, uniform_idx)


def result_checksum(torch, scores, rows):
    """Order-sensitive checksum of a [Q, k] result (device tensors): any N must print the same pair for the same corpus."""
    q, k = rows.shape
    w = (torch.arange(k, device=rows.device, dtype=torch.int64) + 1)[None, :] * (torch.arange(q, device=rows.device, dtype=torch.int64) % 1021 + 1)[:, None]
    r = int(((rows.to(torch.int64) + 1) * w).sum().item() & ((1 << 62) - 1))
    sbits = int((scores.contiguous().view(torch.int32).to(torch.int64) * w).sum().item() & ((1 << 62) - 1))
    return {"rows": r, "score_bits": sbits}


def cpu_baseline_legs(sample, k, n_full, q_one, q_all):
    """The reference's per-query loop restated with its own torch ops (oracle/gip_oracle_torch.py, checked against the numpy
    oracle), timed on the host: (i) ONE thread -- the reference's setting for --batch 1, gip_retrieval.py:255-259 -- and
    (ii) all cores, on a bounded slice of the same corpus; linear extrapolation in the row count."""
    from oracle import gip_oracle_torch as OT
    scv, sci, sqv, sqi = sample
    c32, q32 = scv.astype(np.float32), sqv.astype(np.float32)          # the reference's astype(float32) copies (:268-313), outside its timer too
    import torch
    # threads the process may actually run on: a container can show 256 CPUs and be pinned to a few of them
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = usable
    s1, _ = OT.gip_loop(q32[:q_one], None if sqi is None else sqi[:q_one], c32, sci, k, 1)
    sa, _ = OT.gip_loop(q32[:q_all], None if sqi is None else sqi[:q_all], c32, sci, k, cores)
    threads_seen = int(getattr(OT, "LAST_THREADS", 0)) or cores        # what torch reported INSIDE the loop (gip_loop restores the setting on its way out)
    torch.set_num_threads(max(1, min(cores, 64)))
    scale = n_full / c32.shape[0]
    what = "torch-CPU restatement of gip_retrieval.py:115-126 (mask * corpus -> einsum -> topk, one query at a time, fp32)"
    return {"value": round(1.0 / (s1 * scale), 5), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "%d queries x %d-row slice of the same synthetic corpus, %s, 1 thread (the reference's --batch 1 setting); "
                      "%.3f s/query measured, x%.2f linear extrapolation to %d rows" % (q_one, c32.shape[0], what, s1, scale, n_full),
            "all_cores": {"value": round(1.0 / (sa * scale), 5), "unit": "queries/s", "cores": cores,
                          "sample": "%d queries x %d rows, same loop with torch.set_num_threads(%d); %.3f s/query measured, x%.2f"
                                    % (q_all, c32.shape[0], cores, sa, scale)},
            "host_cpus": os.cpu_count() or 1, "affinity_cpus": usable, "torch_num_threads_in_all_cores_leg": threads_seen,
            "time_budget": "SURVEY.md section 8(d): 32 queries on a 1 M-row slice per leg (~3.5 min for both); --quick runs 4 (1 thread) + 8 (all cores) queries in ~40 s",
            "note": "the reference's op sequence is three elementwise passes over the fp32 corpus per query (mask, product, einsum): one query moves ~15 GB through "
                    "one socket's memory system, and torch parallelises each pass over the rows but the passes stay bandwidth-bound and serial -- the all-cores leg is "
                    "barely faster than one thread, as measured"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="hybrid", choices=["hybrid", "dense", "bm25", "beir"],
                    help="hybrid = config 3 / 4 (768 DLR + 768 dense, 8.84 M rows); dense = config 2 (768 dense, no index array); "
                         "bm25 = config 1 (100 k rows, DLR only, int16 whole-word slice index); beir = config 5 sweep (13 corpus sizes, "
                         "768 DLR + 128 dense), one JSON line per corpus")
    ap.add_argument("--n-docs", type=int, default=0, help="override the workload's corpus size")
    ap.add_argument("--n-queries", type=int, default=0, help="override the workload's query count")
    ap.add_argument("--topk", type=int, default=1000)
    ap.add_argument("--beir-only", default="", help="beir: run this corpus only")
    ap.add_argument("--pq", action="store_true", help="beir: product-quantised first stage (--PQIP: ADC scan to agip_topk candidates, exact GIP rerank) instead of the exact search")
    ap.add_argument("--agip-topk", type=int, default=10000)
    ap.add_argument("--uniform-idx", action="store_true", help="adversarial variant: uniform slice indices")
    ap.add_argument("--cand-cap", type=int, default=0)
    ap.add_argument("--idx-buckets", type=int, default=0)
    ap.add_argument("--sample-period", type=int, default=-1)
    ap.add_argument("--main-chunks", type=int, default=0)
    ap.add_argument("--max-growth", type=int, default=0, help="tuning: max (next sampled chunk) / (rows seen) in 1/16ths")
    ap.add_argument("--aux-cus", type=int, default=-1, help="tuning: CUs the refine/rescoring stream is confined to (0 = no mask)")
    ap.add_argument("--gemm-exclusive", type=int, default=-1)
    ap.add_argument("--overlap-aux", type=int, default=-1)
    ap.add_argument("--progressive-thr", type=int, default=-1, help="tuning: 1 = running exact thresholds only, 2 (library default) = + extrapolated from the scattered fraction seen")
    ap.add_argument("--dense-i8", type=int, default=-1, help="int8 image of the ungated columns in the bound GEMM (dhr_set_option DHR_OPT_DENSE_I8): -1 library default (gated indexes), 0 off, 1 on for dense-only indexes too")
    ap.add_argument("--gemm-variant", type=int, default=-1, help="tuning: bound-GEMM kernel of the fp16 2:4 image (4 = 4 waves of 128 x 128, 5 = 8 waves of 128 x 64: the default)")
    ap.add_argument("--no-progressive-thr", action="store_true", help="A/B: keep the sampled thresholds frozen over the main pass")
    ap.add_argument("--first-rows", type=int, default=0)
    ap.add_argument("--list-stride", type=int, default=0, help="tuning: DHR_PARAM_LIST_STRIDE (uniform slots per query of the bound lists; 262144 = the uniform lists of rounds 3-4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU-baseline slice (SURVEY.md section 8d: 1 M)")
    ap.add_argument("--cpu-queries", type=int, default=32, help="queries of the all-cores CPU leg (SURVEY.md section 8d: 32 queries on a 1 M-row slice; ~3 s per query and M rows on the 256-thread host)")
    ap.add_argument("--cpu-queries-1t", type=int, default=32, help="queries of the one-thread CPU leg (section 8d: 32; ~3.3 s per query and M rows)")
    ap.add_argument("--quick", action="store_true", help="the CPU legs on 8 (all cores) / 4 (one thread) queries instead of section 8d's 32 + 32 (~40 s instead of ~3.5 min)")
    ap.add_argument("--index-path", default="", help="REAL data: the reference's index pickle ([value, index, ids], retrieval/index.py output) or a device-ready index file (GipIndex.save); "
                                                     "with --query-path the timed steps run on it instead of the synthetic corpus (N = 1)")
    ap.add_argument("--query-path", default="", help="REAL data: the reference's query pickle ([value, index, qids])")
    ap.add_argument("--qrels", default="", help="REAL data: qrels (qid 0 docid rel, or the tab-separated layout of retrieval/rcap_eval.py) -> nDCG@10 / MRR@10 / R@1000 of the timed search's lists, attached as `effectiveness`")
    ap.add_argument("--emb-dim", type=int, default=768, help="REAL data: gated columns of the records (the reference's --emb_dim)")
    ap.add_argument("--lamda", type=float, default=1.0, help="REAL data: the reference's --lamda (CLS tail of the queries scaled in fp32)")
    ap.add_argument("--parity-rows", type=int, default=200_000)
    ap.add_argument("--parity-queries", type=int, default=24)
    ap.add_argument("--seed", type=int, default=1237)
    ap.add_argument("--rccl-timeout", type=float, default=300.0, help="N > 1: seconds the bring-up of the library's RCCL communicator (incl. one untimed sharded step) may take before every rank degrades to the host transport")
    ap.add_argument("--two-stage", type=int, default=-1, help="1: after the timed steps also time 3 steps each of the two-stage modes of the reference's docs on the resident index (--theta 0.3 --rerank and --IP --rerank, agip_topk 10000) and attach them as two_stage (default: on for the plain N=1 hybrid invocation)")
    ap.add_argument("--other-configs", type=int, default=-1, help="1: after the headline workload also time 5 steps each of config 2 (dense) and config 1 (bm25) and attach them as other_configs (default: on for the plain N=1 hybrid invocation)")
    ap.add_argument("--data", default="iid", choices=["iid", "clustered"], help="dense columns: iid Gaussian (SURVEY 8d) or the structured variant (2 000 clusters, decaying spectrum, 1 %% near-duplicate rows, 5 %% hot queries)")
    ap.add_argument("--per-step", action="store_true", help="diagnostics: attach the library's own total_ms of every timed step (per_step_total_ms) -- a slow step among fast ones is a host-side stall, not kernel time")
    ap.add_argument("--dist-backend", default="nccl", help="testing only: 'gloo' lets several ranks share ONE GPU (with DHR_BENCH_SINGLE_DEVICE=1)")
    args = ap.parse_args()
    if args.quick:
        args.cpu_queries, args.cpu_queries_1t = min(args.cpu_queries, 8), min(args.cpu_queries_1t, 4)
    if bool(args.index_path) != bool(args.query_path):
        raise SystemExit("--index-path and --query-path go together")

    import torch
    from dhr_amd import _lib, dist as D, synth

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

    ctx = dict(torch=torch, _lib=_lib, D=D, synth=synth, world=world, rank=rank, local_rank=local_rank, device=device)
    if args.workload == "beir":
        for name, n_docs, n_q in BEIR:
            if args.beir_only and name != args.beir_only:
                continue
            spec = dict(name="beir/" + name, n=args.n_docs or n_docs, nq=args.n_queries or n_q, d_dlr=768, d_cls=128, kind="encoder",
                        baseline_config="config 5: DeLADE-CLS-P on BEIR-13 (%s)" % name, seed=args.seed + 5 + (zlib.crc32(name.encode()) % 997))
            out = run_workload(args, spec, ctx)
            if rank == 0:
                print(json.dumps(out), flush=True)
    else:
        specs = {"hybrid": dict(name="hybrid", n=N_MSMARCO, nq=Q_DEV, d_dlr=768, d_cls=768, kind="encoder",
                                baseline_config="config 3: DeLADE-CLS 768+768 dense-hybrid" if world == 1 else "config 4: 768+768 dense-hybrid, corpus row-sharded across %d GPUs" % world),
                 "dense": dict(name="dense", n=N_MSMARCO, nq=Q_DEV, d_dlr=0, d_cls=768, kind="dense", baseline_config="config 2: Aggretriever 768-d dense-only"),
                 "bm25": dict(name="bm25", n=100_000, nq=Q_DEV, d_dlr=768, d_cls=0, kind="bm25",
                              baseline_config="config 1: BM25 densified (DLR only), 100k-passage toy corpus")}
        for sp in specs.values():
            sp["seed"] = args.seed
        spec = specs[args.workload]
        spec["n"] = args.n_docs or spec["n"]
        spec["nq"] = args.n_queries or spec["nq"]
        if args.index_path:
            if world != 1:
                raise SystemExit("the real-data leg runs on one GPU (--gpus 1)")
            spec = load_real(args, torch)
        out = run_workload(args, spec, ctx)
        # configs 1 and 2 on the same clock: the plain default invocation (N = 1, config 3 at full size) also times 5 steps each of the
        # dense-only and the BM25 workload -- same code path, same seed rule, their own roofline and checksum -- and attaches them to the line
        plain = (args.workload == "hybrid" and not args.index_path and world == 1 and not args.n_docs and not args.n_queries and not args.uniform_idx and args.data == "iid"
                 and not args.pq and args.topk == 1000)
        if (args.other_configs == 1 or (args.other_configs < 0 and plain)) and world == 1 and args.workload == "hybrid":
            import copy
            other = {}
            for wl, tag in (("dense", "config2"), ("bm25", "config1")):
                a2 = copy.copy(args)
                a2.workload, a2.steps, a2.warmup, a2.no_cpu_baseline, a2.n_docs, a2.n_queries = wl, 5, 2, True, 0, 0
                a2.two_stage = 0
                o = run_workload(a2, specs[wl], ctx)
                if rank == 0:
                    other[tag] = {key: o[key] for key in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline", "whole_job_frac_of_gemm_roofline",
                                                          "result_checksum", "phase_ms_per_step", "candidates_per_query", "parity_check", "index_device_gb") if key in o}
            if rank == 0:
                out["other_configs"] = other
        if rank == 0:
            print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def load_real(args, torch):
    """The optional real-data leg: an encoded index + query set in the reference's own record layout (tevatron/driver/encode.py:203-204,
    merged by retrieval/index.py) -> the spec run_workload times instead of the synthetic corpus.  The search is the same call; what it adds
    is the second half of BASELINE.json's metric (nDCG@10), computed from the timed search's lists when --qrels is given."""
    from dhr_amd.retrieval import gip_retrieval as G
    q32, qidx, qids = G.load_queries(args.query_path, args.emb_dim, args.lamda)
    if G.GipIndex.is_device_file(args.index_path):
        real = {"device_file": args.index_path}
        probe, docids = G.GipIndex.load(args.index_path)
        n, d_dlr, k_all = probe.n_rows, probe.d_dlr, probe.k
        probe.close()
    else:
        cv, ci, docids, _ = G.load_corpus_shard(args.index_path)
        real = {"cv": cv, "ci": ci}
        n, k_all = cv.shape
        d_dlr = 0 if ci is None else ci.shape[1]
    if qidx is None and d_dlr:
        raise SystemExit("the index has a slice-index array, the queries have none")
    real.update(q32=q32, qi=qidx, qids=[str(x) for x in qids], docids=[str(x) for x in docids])
    return dict(name="real", n=int(n), nq=int(q32.shape[0]), d_dlr=int(d_dlr), d_cls=int(k_all - d_dlr), kind="real", seed=args.seed, real=real,
                baseline_config="real data: index %s, queries %s" % (os.path.basename(args.index_path), os.path.basename(args.query_path)))


def run_workload(args, spec, ctx):
    torch, _lib, D, synth = ctx["torch"], ctx["_lib"], ctx["D"], ctx["synth"]
    world, rank, local_rank, device = ctx["world"], ctx["rank"], ctx["local_rank"], ctx["device"]
    from dhr_amd.retrieval.gip_retrieval import GipIndex
    d_dlr, d_cls, n, nq, k = spec["d_dlr"], spec["d_cls"], spec["n"], spec["nq"], args.topk
    K = d_dlr + d_cls
    lo, hi = D.shard_bounds(n, world, rank)
    seed = spec["seed"]

    # ---------------- data (outside the timed region)
    t_gen = time.perf_counter()
    real = spec.get("real")
    if real is not None:
        qv = torch.from_numpy(real["q32"]).to(device)
        qi = None if real["qi"] is None else torch.from_numpy(real["qi"]).to(device)
        cv = ci = None
        if "cv" in real:
            cv = torch.from_numpy(np.ascontiguousarray(real["cv"])).to(device)
            ci = None if real["ci"] is None else torch.from_numpy(np.ascontiguousarray(real["ci"])).to(device)
    elif spec["kind"] == "bm25":
        # config 1: whole-word vocabulary (2.6 M terms -> int16 slice index < 3400), no background, integer query weights;
        # generated on the host (the numpy generator covers int16 indices), rows [lo, hi) of the SAME matrix on every rank
        cvh, cih, qvh, qih = synth.make_pair(seed, n, nq, d_dlr, 0, kind="bm25")
        cv, ci = torch.from_numpy(cvh[lo:hi]).to(device), torch.from_numpy(cih[lo:hi]).to(device)
        qv, qi = torch.from_numpy(qvh).to(device), torch.from_numpy(qih).to(device)
        del cvh, cih, qvh, qih
    else:
        cm = synth.torch_cluster_model(seed, d_cls, device) if (args.data == "clustered" and d_cls) else None
        cv, ci = gen_rows(torch, synth, device, seed, lo, hi, d_dlr, d_cls, 30, 90, args.uniform_idx, clustered=cm, dup=True)
        qv, qi = gen_rows(torch, synth, device, seed + 999_983, 0, nq, d_dlr, d_cls, 4, 12, args.uniform_idx, clustered=cm, hot_frac=synth.HOT_FRAC)
        del cm
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    t_build = time.perf_counter()
    _lib.check(_lib.load().dhr_set_option(_lib.OPT_DENSE_I8, args.dense_i8), "dhr_set_option")
    if real is not None and cv is None:
        index, _ = GipIndex.load(real["device_file"], device=local_rank)
    else:
        index = GipIndex(cv, ci, device=local_rank, row_offset=lo, idx_buckets=args.idx_buckets)
    dense_i8 = bool(index.info(_lib.INFO_DENSE_I8))
    gated_i8 = bool(index.info(_lib.INFO_GATED_I8))
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t_build
    index.set_param(_lib.PARAM_PROFILE, 1)
    for flag, prm in ((args.cand_cap, _lib.PARAM_CAND_CAP), (args.max_growth, _lib.PARAM_MAX_GROWTH), (args.main_chunks, _lib.PARAM_MAIN_CHUNKS),
                      (args.first_rows, _lib.PARAM_FIRST_ROWS), (args.list_stride, _lib.PARAM_LIST_STRIDE)):
        if flag:
            index.set_param(prm, flag)
    for flag, prm in ((args.sample_period, _lib.PARAM_SAMPLE_PERIOD), (args.aux_cus, _lib.PARAM_AUX_CUS), (args.overlap_aux, _lib.PARAM_OVERLAP_AUX),
                      (args.gemm_variant, _lib.PARAM_GEMM_VARIANT), (args.gemm_exclusive, _lib.PARAM_GEMM_EXCLUSIVE)):
        if flag >= 0:
            index.set_param(prm, flag)
    if args.no_progressive_thr:
        index.set_param(_lib.PARAM_PROGRESSIVE_THR, 0)
    elif args.progressive_thr >= 0:
        index.set_param(_lib.PARAM_PROGRESSIVE_THR, args.progressive_thr)

    # host samples for the CPU baseline and the in-bench parity check (rank 0, N=1 only; BEFORE the corpus tensors are dropped)
    cpu_sample = par_sample = None
    if rank == 0 and world == 1 and cv is not None:
        def host_slice(m, mq):
            m, mq = min(m, hi - lo), min(mq, nq)
            return (cv[:m].cpu().numpy(), None if ci is None else ci[:m].cpu().numpy(), qv[:mq].cpu().numpy(), None if qi is None else qi[:mq].cpu().numpy())
        if not args.no_cpu_baseline:
            cpu_sample = host_slice(args.cpu_rows, max(args.cpu_queries, args.cpu_queries_1t))
        par_sample = host_slice(args.parity_rows, args.parity_queries)
    pq = None
    t_pq = 0.0
    if args.pq:
        # --PQIP (gip_retrieval.py:167-231): product-quantised first stage over the whole vector (M = 64 sub-quantisers, 8 bits:
        # quantize_index.py:29) -> agip_topk candidates -> exact GIP rerank.  Codebooks: the library's own k-means (parity with faiss unpinned).
        from dhr_amd.retrieval import quantize_index as QI
        t_pq = time.perf_counter()
        # (N > 1: every rank trains on ITS rows with the same deterministic seeding; the sharded search needs ONE set of codebooks, so rank 0's
        # are broadcast and every rank encodes its shard with them)
        cb, codes, _ = QI.train_and_encode(cv, 64, 8, iters=10, device=local_rank)
        if world > 1:
            import torch.distributed as dist
            cb_t = (cb if hasattr(cb, "data_ptr") else torch.from_numpy(np.asarray(cb))).to(device).contiguous()
            dist.broadcast(cb_t, src=0)
            cb = cb_t
            codes = QI.encode(cv, cb, device=local_rank)
        pq = QI.PqIndex(cb, codes, nbits=8, device=local_rank, row_offset=lo)
        del cb, codes
        torch.cuda.synchronize()
        t_pq = time.perf_counter() - t_pq
    del cv, ci
    torch.cuda.empty_cache()
    k1 = min(args.agip_topk, hi - lo)

    def step():
        if pq is not None:
            if world > 1:      # config 5 literally: PQ first stage x row shards -- global agip_topk cut, per-shard exact rerank, all-gather + rank merge
                return D.pq_sharded_search(pq, index, qv, qi, args.agip_topk, min(k, n), n_total=n)
            s1, r1 = pq.search(qv, k1, out_device=True)                       # ADC scan: top-agip_topk by the quantised inner product
            return D.rerank_topk(index, qv, qi, r1, min(k, k1))               # exact gated inner product of the candidates (:205-215), top-k
        if world > 1:
            return D.sharded_search(index, qv, qi, k)        # common thresholds + one all-gather of the shard lists
        return index.search(qv, qi, min(k, n), out_device=True)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: the communicator is brought up under a watchdog (dhr_amd.dist.bring_up): RCCL id + ncclCommInitRank + ONE untimed sharded
    # step run in a worker thread; this thread waits at most --rccl-timeout seconds, then the ranks vote over a gloo control group.  One
    # "no" (an exception, or a rank still blocked because a peer never arrived) and EVERY rank drops RCCL and takes torch.distributed
    # all-gathers on host buffers (dhr_comm_create_callback: the SAME control flow in the library, only the gathers change); the JSON line
    # says which transport the timed steps used and how many ranks RCCL itself reports (ncclCommCount).
    sharded_impl = None
    comm = None
    if world > 1 and pq is None:
        import torch.distributed as dist
        want = "host" if (dist.get_backend() != "nccl" and os.environ.get("DHR_BENCH_FORCE_RCCL_TRIAL") != "1") or os.environ.get("DHR_SHARDED_TRANSPORT") == "host" else "rccl"

        def trial(c):
            D.sharded_search(index, qv, qi, k, comm=c)
            torch.cuda.synchronize(device)
        comm = D.checked_comm(index, None, trial, args.rccl_timeout, want)
        sharded_impl = "dhr_search_sharded: " + comm.note
        if comm.transport != "rccl" and want == "rccl":
            print("[bench] rank %d: %s" % (rank, sharded_impl), file=sys.stderr)
    kk = min(k, n)
    host_s = torch.empty((nq, kk), dtype=torch.float32).pin_memory()
    host_r = torch.empty((nq, kk), dtype=torch.int64).pin_memory()

    copy_stream = torch.cuda.Stream(device=device)

    def step_d2h():
        """One step as the reference's timer sees it (gip_retrieval.py:107,161-163): the query loop incl. top-k AND the copy of the
        result lists to the host (rank 0 holds the merged lists).  The search returns with its lists complete (its last action is a host
        read), so their copy goes out on a second stream and runs beside the NEXT batch's search -- the 84 MB of a batch's lists are
        1.5 ms of PCIe; the timed region still ends only when the last batch's lists are on the host (the closing synchronize)."""
        gs, gr = step()
        if rank == 0:
            copy_stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(copy_stream):
                host_s.copy_(gs[:, :kk], non_blocking=True)
                host_r.copy_(gr[:, :kk], non_blocking=True)
            gs.record_stream(copy_stream)
            gr.record_stream(copy_stream)
        return gs, gr

    # N > 1: the kernel timers of a sharded step cost it a host synchronisation between its stages (begin | exchange | first slice | exchange |
    # rest), so the timed steps run WITHOUT them and the same number of extra, untimed steps below collects the kernel times and list statistics
    stats_in_timed = world == 1 or pq is not None
    if not stats_in_timed:
        index.set_param(_lib.PARAM_PROFILE, 0)
    for _ in range(args.warmup):
        step_d2h()
    acc = {"gemm_ms": 0.0, "gemm_flops_alg": 0.0, "launches": 0}
    stats_acc = {}

    def collect():
        st = index.stats() if pq is None else dict.fromkeys(("gemm_ms", "phases", "gemm_flops_alg", "refine_ms", "rescore_ms", "select_ms", "prep_ms", "total_ms",
                                                              "candidates_bound", "candidates_exact", "overflow_retries", "gemm_rows", "sample_fallback_queries"), 0)
        if pq is not None:
            ms_scan, by_scan = pq.last_scan()
            stats_acc["adc_scan_ms"] = stats_acc.get("adc_scan_ms", 0) + ms_scan
            stats_acc["adc_code_bytes"] = stats_acc.get("adc_code_bytes", 0) + by_scan
        acc["gemm_ms"] += st["gemm_ms"]
        acc.setdefault("total_by_step", []).append(round(st["total_ms"], 3))
        acc["launches"] += st["phases"]
        acc["gemm_flops_alg"] += st["gemm_flops_alg"]           # algorithmic: real Q and K of every launch
        for key in ("refine_ms", "rescore_ms", "select_ms", "prep_ms", "total_ms", "candidates_bound", "candidates_exact",
                    "overflow_retries", "gemm_rows", "sample_fallback_queries"):
            stats_acc[key] = stats_acc.get(key, 0) + st[key]

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gs, gr = step_d2h()
        if stats_in_timed:
            collect()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed * 1e3 / args.steps
    qps = nq * args.steps / elapsed

    checksum = result_checksum(torch, gs[:, :kk], gr[:, :kk])
    # side figure: the same K steps with the lists left in device memory (what a caller that consumes them on the GPU sees)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    ms_device_resident = (time.perf_counter() - t1) * 1e3 / args.steps
    if not stats_in_timed:
        index.set_param(_lib.PARAM_PROFILE, 1)
        for _ in range(args.steps):
            step()
            collect()
        barrier()
    gemm_ms, gemm_flops_alg, launches = acc["gemm_ms"], acc["gemm_flops_alg"], acc["launches"]

    # side figure 2 (N = 1): the bound GEMM ALONE.  In the timed steps refine / rescoring / select of a chunk run beside the next chunk's GEMM
    # (library default since round 4), so the GEMM's launch durations there include what it loses to the gathers; two extra steps with
    # DHR_PARAM_OVERLAP_AUX = 0 time the same launches with the chip to themselves (roofline.achieved; the timed region's own figure beside it)
    serial = None
    if world == 1 and pq is None and args.overlap_aux < 0:
        index.set_param(_lib.PARAM_OVERLAP_AUX, 0)
        step()
        s_ms = s_fl = 0.0
        s_n = 0
        t2 = time.perf_counter()
        for _ in range(3):
            step()
            st = index.stats()
            s_ms += st["gemm_ms"]; s_fl += st["gemm_flops_alg"]; s_n += st["phases"]
        torch.cuda.synchronize()
        serial = {"gemm_ms_per_step": s_ms / 3, "tf": s_fl / (s_ms * 1e-3) / 1e12 if s_ms > 0 else 0.0, "launches": s_n,
                  "ms_per_step": (time.perf_counter() - t2) * 1e3 / 3}
        index.set_param(_lib.PARAM_OVERLAP_AUX, -1)

    # side figure 3 (N = 1, hybrid): the two-stage modes every doc page of the reference recommends (gip_retrieval.py:128-156;
    # docs/dhr/msmarco-passage-train-eval.md:114-125), both stages on the device (dhr_search_rerank), agip_topk 10 000 -> top-k, result lists on the host
    index_device_bytes = index.device_bytes()          # (before the two-stage leg: agip_topk 10 000 grows the workspace)
    two_stage = None
    plain_hybrid = (spec["name"] == "hybrid" and world == 1 and pq is None and not args.n_docs and not args.n_queries and not args.uniform_idx and args.topk == 1000)
    if spec["name"] == "hybrid" and world == 1 and pq is None and (args.two_stage == 1 or (args.two_stage < 0 and plain_hybrid and args.data == "iid")):
        q32 = qv.float()
        k1 = min(args.agip_topk, n)
        two_stage = {"agip_topk": k1, "topk": min(k, k1), "steps": 3}
        for name, qa, qia in (("theta0.3_rerank", torch.where(q32 > 0.3, q32, torch.zeros_like(q32)), qi), ("ip_rerank", q32, None)):
            index.search_rerank(qa, qia, q32, qi, k1, min(k, k1))
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            for _ in range(3):
                s2, r2 = index.search_rerank(qa, qia, q32, qi, k1, min(k, k1))
            torch.cuda.synchronize()
            two_stage[name] = {"ms_per_step": round((time.perf_counter() - t3) * 1e3 / 3, 2), "queries_per_s": round(nq * 3 / (time.perf_counter() - t3), 1),
                               "result_checksum": {"rows": int(((r2.astype(np.int64) + 1) * (np.arange(r2.shape[1], dtype=np.int64) + 1)[None, :]).sum() & ((1 << 62) - 1))}}
        del q32

    # ---- N > 1: size-independent checks of the sharded result, outside the timed region (the oracle cannot hold the
    # corpus): sorted lists of distinct rows; every returned score is the exact score of its row on the rank that holds
    # the row (dhr_score_rows, an independent code path); no sampled row of any shard outside a list beats its k-th score
    dist_check = None
    if world > 1:
        import torch.distributed as dist
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
        ach_k = serial["tf"] if serial else ach_tf
        sparse_layout = d_dlr > 0 and args.idx_buckets in (0, 2)
        stage_layout = sparse_layout or (d_dlr == 0 and args.idx_buckets == 0)      # dense-only indexes run on the stage images too (ts = 0)
        variant = args.gemm_variant if args.gemm_variant >= 0 else 5
        # the kernel the timed launches really ran, asked of the library (DHR_INFO_GEMM_KERNEL) -- until round 5 this label was re-derived here
        # from the index flags and named gemm_filter_wx_kernel for dense-only int8 indexes, which run gemm_filter_g8_kernel (DHR_DENSE_G8)
        kid = int(index.info(_lib.INFO_GEMM_KERNEL))
        g8_ops = ("gated half on v_smfmac_i32_32x32x64_i8, ungated half on v_mfma_i32_32x32x32_i8" if d_dlr and d_cls else
                  "v_smfmac_i32_32x32x64_i8 (2:4 int8)" if d_dlr else "v_mfma_i32_32x32x32_i8 on the int8 stage images, no gated stage")
        kernel = {3: "gemm_filter_wx_kernel<NI=2> (bound GEMM on the %s + fused threshold filter, 8 waves)" % ("2:4 sparse matrix cores" if sparse_layout else "matrix cores, 32-column stage images"),
                  4: "gemm_filter_wx_kernel<NI=4> (bound GEMM on the %s + fused threshold filter, 4 waves)" % ("2:4 sparse matrix cores" if sparse_layout else "matrix cores, 32-column stage images"),
                  5: "gemm_filter_g8_kernel (integer bound GEMM: %s, fused threshold filter, 8 waves)" % g8_ops,
                  6: "gemm_filter_g8p_kernel (integer bound GEMM with persistent workgroups: %s)" % g8_ops}.get(kid, "unknown (DHR_INFO_GEMM_KERNEL = %d)" % kid)
        # fabric-side (Infinity Cache + HBM) read bytes: PMC passes cannot run inside this process (a torch process hangs under
        # --pmc), so `traffic` is the per-corpus-row figure of the committed rocprofv3 FETCH_SIZE pass over the SAME kernel
        # (tools/prof.sh -> profiles/r02_gemm_pmc.txt; torch-free driver, 6 980 queries) x the average rows per launch
        traffic_per_row = TRAFFIC_BYTES_PER_ROW.get((d_dlr, d_cls, nq, kid, int(dense_i8), int(gated_i8))) if stage_layout and world == 1 else None
        # `peak`: the contract's figure for this metric -- the dense fp16 / bf16 matrix peak the north-star prices the Q x D^T against
        # (BASELINE.json).  Beside it the peak of the instruction mix the kernel really issues: gated columns on the fp16 2:4
        # instruction (two bucket columns per slice at twice the rate = the fp16 peak per algorithmic column) or, gated_i8, on the
        # int8 2:4 instruction (6 650 logical TOP/s measured with zero operands, profiles/r03_smfmac8_probe.txt = 3 325 per
        # algorithmic column); ungated columns on fp16 or, dense_i8, on the dense int8 instruction
        peak = MFMA_PEAK_TFLOPS
        peak_mix = (d_dlr + d_cls) / (d_dlr / (SMFMAC_I8_PEAK_TOPS / 2 if gated_i8 else MFMA_PEAK_TFLOPS) + d_cls / (MFMA_PEAK_I8_TOPS if dense_i8 else MFMA_PEAK_TFLOPS))
        rows_per_launch = stats_acc.get("gemm_rows", 0) / max(launches, 1)
        alg_bytes_per_row = 2 * K + d_dlr * (2 if spec["kind"] == "bm25" else 1)
        out = {
            "metric": "queries/sec (exact top-%d, brute-force dense-hybrid GIP retrieval)" % k,
            "value": round(qps, 2), "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "timed_region": "K steps, each: all queries against the whole corpus, exact top-k, result lists copied to pinned host memory (the reference's timer, "
                            "gip_retrieval.py:107,161-163; a batch's copy runs on a second stream beside the next batch's search, the region ends when the last "
                            "copy has landed); corpus index and queries resident in HBM",
            "dtype": ("i8 (bound GEMM: gated AND ungated columns int8 x int8 -> int32 on the matrix cores, every gated rounding upward, the ungated quantisation error paid by the "
                      "filter margin) + f64-accumulated f16 x f32 exact rescoring of the survivors -- the returned scores are the exact scores rounded once, identical to the fp16 bound's" if gated_i8 else
                      "f16 + i8 (bound: gated columns fp16 x fp16 -> fp32, ungated columns int8 x int8 -> int32 on the matrix cores, the quantisation error paid by the filter margin; "
                      "fp64-accumulated exact rescoring of the survivors in fp16 x fp32 -- results identical to the fp16 bound)" if dense_i8 else
                      "f16 (fp16 x fp16 -> fp32 on the matrix cores for the bound; fp64-accumulated exact rescoring)"),
            "data": "synthetic" if real is None else "real (index %s, queries %s)" % (args.index_path, args.query_path),
            "config": {"workload": ("%s: %d rows x (%d DLR + %d dense) fp16%s, %d queries, top-%d, %s"
                                    % (spec["name"], n, d_dlr, d_cls,
                                       (" + int16 slice index" if spec["kind"] == "bm25" else " + uint8 slice index") if d_dlr else "", nq, k,
                                       "uniform slice index (adversarial)" if args.uniform_idx else
                                       "whole-word vocabulary, no background" if spec["kind"] == "bm25" else "densify-rule slice index"))
                                   + ("; dense columns CLUSTERED (2 000 clusters, sigma_within 0.3, decaying spectrum, 1 % near-duplicate rows, 5 % hot queries)" if args.data == "clustered" and d_cls else ""),
                       "baseline_config": spec["baseline_config"],
                       "parallelism": "rowshard%d+allgather" % world if world > 1 else "1gpu",
                       **({"collectives": sharded_impl} if sharded_impl else {})},
            **({"n_ranks_seen_by_rccl": (comm.ranks_seen() if comm.transport == "rccl" else 0), "sharded_transport": comm.transport} if comm is not None else {}),
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": round(ach_tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach_tf / peak, 4),
                         "achieved_note": ("hipEvent durations of rank 0's launches in %d extra, untimed steps of the same sharded search (the timers cost a sharded step a host "
                                           "synchronisation between its stages, so the timed steps run without them)" % args.steps) if not stats_in_timed else
                                          "algorithmic flops / hipEvent durations of the kernel's launches INSIDE the timed region (where they share the chip with the previous "
                                          "chunk's refine / rescoring / select)",
                         "achieved_kernel_alone": round(ach_k, 1) if serial else None, "frac_kernel_alone": round(ach_k / peak, 4) if serial else None,
                         "kernel_alone_note": ("hipEvent durations of the same %d launches per step in 3 extra steps with refine / rescoring / select serialised behind each "
                                               "launch (DHR_PARAM_OVERLAP_AUX = 0, %.1f ms per step)" % (serial["launches"] // 3, serial["ms_per_step"])) if serial else None,
                         "peak_note": "dense fp16 / bf16 matrix peak (MI355X_MICROARCH.md): the roofline BASELINE.json prices this metric against; algorithmic flops 2 x Q x rows x (d_dlr + d_cls)",
                         "peak_instruction_mix": round(peak_mix, 1), "frac_of_instruction_mix_peak": round(ach_tf / peak_mix, 4),
                         "frac_of_instruction_mix_peak_kernel_alone": round(ach_k / peak_mix, 4) if serial else None,
                         "traffic": None if traffic_per_row is None else round(traffic_per_row * rows_per_launch, 0),
                         "traffic_unit": "bytes per launch (rocprofv3 --pmc FETCH_SIZE, gfx950-corrected; profiles/r06_gemm_pmc.txt; fp16-gated hybrid images: r02_gemm_pmc.txt)",
                         "algorithmic_bytes_per_launch": round(alg_bytes_per_row * rows_per_launch, 0),
                         "launches": launches, "avg_launch_ms": round(gemm_ms / max(launches, 1), 3),
                         "avg_launch_ms_kernel_alone": round(serial["gemm_ms_per_step"] * 3 / max(serial["launches"], 1), 3) if serial else None,
                         "alg_flops_per_step": gemm_flops_alg / args.steps},
            "whole_job_frac_of_gemm_roofline": round(qps * 2.0 * n * K / (peak * 1e12 * world), 4),
            "device_resident": {"ms_per_step": round(ms_device_resident, 3), "queries_per_s": round(nq / (ms_device_resident * 1e-3), 1),
                                "note": "the same steps with the result lists left in device memory"},
            "result_checksum": checksum,
            "phase_ms_per_step": {key: round(v / args.steps, 3) for key, v in stats_acc.items() if key.endswith("_ms")}
                                 | {"gemm_ms": round(gemm_ms / args.steps, 3)},
            "candidates_per_query": {"bound": round(stats_acc["candidates_bound"] / args.steps / nq, 1),
                                     "exact": round(stats_acc["candidates_exact"] / args.steps / nq, 1)},
            **({"per_step_total_ms": acc.get("total_by_step", [])} if args.per_step else {}),
            "sample_fallback_queries_per_step": stats_acc["sample_fallback_queries"] / args.steps,
            "overflow_retries_per_step": stats_acc["overflow_retries"] / args.steps,
            "setup_s": {"generate": round(t_gen, 2), "index_build": round(t_build, 2)},
            "index_device_gb": round(index_device_bytes / 1e9, 2),
        }
        if two_stage is not None:
            out["two_stage"] = two_stage
        if real is not None and args.qrels:
            # the second half of BASELINE.json's metric on real data: nDCG@10 (+ MRR@10, R@1000) of the lists the timed steps produced
            from dhr_amd.retrieval import trec
            rows_h, sc_h = host_r.numpy(), host_s.numpy()
            run = {}
            for qx, qid in enumerate(real["qids"]):
                keep = [(real["docids"][int(r)], float(sc_h[qx, j])) for j, r in enumerate(rows_h[qx]) if r >= 0 and real["docids"][int(r)] != qid]   # gip_retrieval.py:340
                run[qid] = ([d for d, _ in keep], [x for _, x in keep])
            out["effectiveness"] = dict(trec.effectiveness(trec.read_qrels_any(args.qrels), run), qrels=args.qrels,
                                        note="trec_eval's definitions (dhr_amd/retrieval/trec.py effectiveness); the docid == query_id filter of gip_retrieval.py:340 applied")
        if pq is not None:
            # the dominant kernel of this mode is the ADC scan: HBM / LDS-gather bound integer-index work (roofline on HBM bytes)
            # Two rooflines, honestly labelled.  HBM: the UNIQUE code bytes of a step are rows x 64 B -- every query pair re-reads them, but from
            # L2 / the Infinity Cache, so the HBM figure is tiny and NOT what bounds the scan.  What does: the data-indexed table reads, 64
            # ds_read_b64 per (row, query pair) at ~3.5-way bank conflicts; their conflict-free rate is 32 lanes per CU and clock
            # (MI355X_MICROARCH.md LDS table: ds_read_b64 = 2 cycles per wave instruction) = 256 CUs x 32 x 2.4 GHz = 19.7 T reads / s.
            scan_s = stats_acc["adc_scan_ms"] * 1e-3
            unique_bytes = (hi - lo) * 64.0 * args.steps
            gbs = unique_bytes / scan_s / 1e9 if scan_s > 0 else 0.0
            pairs = (nq + 1) // 2
            lookups = (hi - lo) * 64.0 * pairs * args.steps
            lds_peak = 256 * 32 * 2.4e9
            out["roofline"] = {"bound": "hbm", "kernel": "adc_scan_kernel (product-quantised inner product: 64 B of codes per row, fp32 lookup tables of a query pair in LDS, fused threshold filter)",
                               "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "traffic": None,
                               "algorithmic_bytes_per_step": unique_bytes / args.steps,
                               "note": "UNIQUE code bytes (rows x 64 B per step) / scan time: the scan is NOT HBM-bound -- every query pair re-reads the codes from L2 / the "
                                       "Infinity Cache (%.0f GB of code reads per step at %.0f GB/s); its bound is the LDS gather below" %
                                       (stats_acc["adc_code_bytes"] / args.steps / 1e9, stats_acc["adc_code_bytes"] / scan_s / 1e9 if scan_s > 0 else 0.0),
                               "lds_gather": {"achieved": round(lookups / scan_s / 1e12, 2) if scan_s > 0 else 0.0, "peak": round(lds_peak / 1e12, 2), "unit": "T table reads/s (ds_read_b64 lanes)",
                                              "frac": round(lookups / scan_s / lds_peak, 4) if scan_s > 0 else 0.0,
                                              "note": "rows x 64 sub-quantisers x query pairs data-indexed LDS reads per step; conflict-free peak 32 lanes per CU and clock"}}
            out["config"]["first_stage"] = "PQ M=64 nbits=8 ADC scan, agip_topk %d, exact GIP rerank of the candidates" % k1
            out["pq"] = {"index_device_mb": round(pq.device_bytes() / 1e6, 1), "train_encode_s": round(t_pq, 2), "adc_scan_ms_per_step": round(stats_acc["adc_scan_ms"] / args.steps, 3)}
            pq.close()
        if cpu_sample is not None:
            out["cpu_baseline"] = cpu_baseline_legs(cpu_sample, k, n, min(args.cpu_queries_1t, nq), min(args.cpu_queries, nq))
        if par_sample is not None:
            # parity of the HIP path on a slice of the same corpus against the oracle (outside the timed region)
            from oracle import gip_oracle as O
            scv, sci, sqv, sqi = par_sample
            six = GipIndex(scv, sci, device=local_rank)
            kp = min(k, scv.shape[0])
            gs2, gr2 = six.search(sqv.astype(np.float32), sqi, kp)
            six.close()
            bad = 0
            c32 = scv.astype(np.float32)
            for i in range(sqv.shape[0]):
                ex = O.gip_scores_f64(sqv[i].astype(np.float32), None if sqi is None else sqi[i], c32, sci)
                try:
                    O.check_topk(gr2[i], gs2[i], ex, kp)
                except AssertionError:
                    bad += 1
            out["parity_check"] = {"queries": int(sqv.shape[0]), "rows": int(scv.shape[0]), "failed": bad}
        if dist_check is not None:
            out["parity_check"] = dist_check
    index.close()
    del index, qv, qi
    torch.cuda.empty_cache()
    return out


# fabric-side read bytes per corpus row of the bound GEMM, from the committed PMC pass (d_dlr, d_cls, queries, kernel variant)
# (d_dlr, d_cls, queries, kernel variant, dense_i8) -> fabric-side read bytes per corpus row, profiles/r02_gemm_pmc.txt
# (key: d_dlr, d_cls, queries, DHR_INFO_GEMM_KERNEL id, dense_i8, gated_i8)
TRAFFIC_BYTES_PER_ROW = {(768, 768, 6980, 3, 0, 0): 34.6e3,     # gemm_filter_wx_kernel, fp16 image of the ungated columns: 17.31 GB per 500 000-row launch (r02)
                         (768, 768, 6980, 3, 1, 0): 26.2e3,     # ... int8 image of the ungated columns, fp16 gated: 13.10 GB (r02)
                         (768, 768, 6980, 5, 1, 1): 22.5e3,     # gemm_filter_g8_kernel, gated_i8 (default since round 3): 11.23 GB (profiles/r06_gemm_pmc.txt: 11.24; r05 / r04: 11.23, r03: 11.27)
                         (0, 768, 6980, 3, 0, 0): 14.6e3,       # dense-only index, fp16 stage images: 7.30 GB (r04)
                         (0, 768, 6980, 3, 1, 0): 6.06e3,       # dense-only index, int8 stage images on gemm_filter_wx_kernel (DHR_DENSE_G8=0): 3.03 GB (r04)
                         (0, 768, 6980, 5, 1, 0): 6.06e3}       # ... on gemm_filter_g8_kernel with no gated stage (default since round 5): the same operand images and tile order


if __name__ == "__main__":
    main()
