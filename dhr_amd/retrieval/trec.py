"""Host-side TREC run helpers around the search path: the file-based shard reduce and the
capped-recall check.

Mirrors the command lines of
  retrieval/merge.result.py:13-42  (result00.trec .. resultNN.trec -> result.trec)
  retrieval/rcap_eval.py:4-29      (R_cap@cutoff of a run against tab-separated qrels)
  retrieval/evaluation/custom_metrics.py:34-58 (the capped-recall definition)

Both are plain text processing on the host; the in-process shard reduce (no text files) is
dhr_amd.dist / dhr_merge_topk.  The file merge keeps the reference's semantics -- per query,
concatenate the shard lists in shard order, keep the `topk` best by score, re-rank from 1 -- with
one deliberate difference: equal scores keep shard order (a stable sort), where the reference's
reversed argsort leaves the order of ties unspecified.
"""
import argparse
import os

import numpy as np


def read_run(path):
    """Parse a TREC run file into {qid: ([docid...], [score...])}, preserving line order."""
    run = {}
    with open(path, 'r') as f:
        for lineno, line in enumerate(f, 1):
            fields = line.split()
            if not fields:
                continue
            if len(fields) != 6:
                raise ValueError('{}:{}: expected 6 fields, got {}'.format(path, lineno, len(fields)))
            docs, scores = run.setdefault(fields[0], ([], []))
            docs.append(fields[2])
            scores.append(float(fields[4]))
    return run


def merge_runs(runs, topk):
    """Reduce per-shard runs ({qid: (docids, scores)}) to the global top-`topk` per query.

    Queries keep their order of first appearance (merge.result.py:22-29 fills a dict the same way).
    """
    pooled = {}
    for run in runs:
        for qid, (docs, scores) in run.items():
            d, s = pooled.setdefault(qid, ([], []))
            d.extend(docs)
            s.extend(scores)
    merged = {}
    for qid, (docs, scores) in pooled.items():
        order = np.argsort(-np.asarray(scores, dtype=np.float64), kind='stable')[:topk]
        merged[qid] = ([docs[i] for i in order], [scores[i] for i in order])
    return merged


def write_run(path, run, run_name):
    with open(path, 'w') as fout:
        for qid, (docs, scores) in run.items():
            for rank, (docid, score) in enumerate(zip(docs, scores), 1):
                fout.write('{} Q0 {} {} {} {}\n'.format(qid, docid, rank, score, run_name))


def merge_main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--total_shrad", type=int, default=1)
    parser.add_argument("--topk", type=int, default=1000)
    parser.add_argument("--run_name", default='dhr')
    parser.add_argument("--dir", default='.', help="directory holding resultNN.trec (default: cwd, as the reference)")
    args = parser.parse_args(argv)
    runs = [read_run(os.path.join(args.dir, 'result{:02d}.trec'.format(s))) for s in range(args.total_shrad)]
    print('write results ...')
    write_run(os.path.join(args.dir, 'result.trec'), merge_runs(runs, args.topk), args.run_name)


def read_qrels(path):
    """qid \\t _ \\t docid \\t rel  ->  {qid: {docid: rel}} (rcap_eval.py:11-18)."""
    qrels = {}
    with open(path, 'r') as f:
        for line in f:
            if not line.strip():
                continue
            qid, _, docid, rel = line.strip().split('\t')
            qrels.setdefault(qid, {})[docid] = int(rel)
    return qrels


def recall_cap(qrels, results, k_values):
    """Capped recall: per query, relevant docs among the k best / min(#relevant, k); summed over the
    queries of the run, divided by the number of qrel queries, rounded to 5 places
    (custom_metrics.py:46-55).  `results` is {qid: {docid: score}}."""
    totals = {k: 0.0 for k in k_values}
    k_max = max(k_values)
    for qid, doc_scores in results.items():
        judged = qrels[qid]
        n_rel = sum(1 for rel in judged.values() if rel > 0)
        ranked = sorted(doc_scores.items(), key=lambda kv: kv[1], reverse=True)[:k_max]
        hits = np.cumsum([judged.get(docid, 0) > 0 for docid, _ in ranked]) if ranked else np.zeros(0, int)
        for k in k_values:
            found = int(hits[min(k, len(hits)) - 1]) if len(hits) else 0
            totals[k] += found / min(n_rel, k)
    return {'R_cap@{}'.format(k): round(totals[k] / len(qrels), 5) for k in k_values}


def read_qrels_any(path):
    """Qrels in either of the two layouts the reference's pipelines use: the tab-separated 4 columns of rcap_eval.py:11-18, or the
    space-separated TREC layout `qid 0 docid rel` that trec_eval reads (docs/dhr/msmarco-passage-train-eval.md:153-154)."""
    qrels = {}
    with open(path, 'r') as f:
        for line in f:
            fields = line.split()
            if not fields:
                continue
            if len(fields) != 4:
                raise ValueError('{}: expected 4 fields per line, got {!r}'.format(path, line))
            qrels.setdefault(fields[0], {})[fields[2]] = int(fields[3])
    return qrels


def effectiveness(qrels, run, ndcg_k=10, mrr_k=10, recall_k=1000):
    """The second half of BASELINE.json's metric: nDCG@10 (plus MRR@10 and R@1000, what the reference's doc pages report with trec_eval:
    docs/dhr/msmarco-passage-train-eval.md:153-154) of a run {qid: ([docid...], [score...])} in rank order.  trec_eval's definitions:
    DCG = sum over the ranks i = 1..k of rel_i / log2(i + 1) with the judged gain as it stands, the ideal DCG from the query's judged
    gains sorted descending; MRR = 1 / rank of the first doc with rel > 0 among the first k; recall = relevant found among the first k /
    all relevant.  Averages run over the queries of the run that have at least one relevant judgment (trec_eval skips the others);
    docs that are not judged count as rel = 0."""
    disc = 1.0 / np.log2(np.arange(2, max(ndcg_k, 2) + 2))
    n = 0
    ndcg = mrr = rec = 0.0
    for qid, (docs, _scores) in run.items():
        judged = qrels.get(qid)
        if not judged:
            continue
        rels = sorted((r for r in judged.values() if r > 0), reverse=True)
        if not rels:
            continue
        n += 1
        gains = np.array([max(judged.get(d, 0), 0) for d in docs[:ndcg_k]], dtype=np.float64)
        ideal = np.array(rels[:ndcg_k], dtype=np.float64)
        ndcg += float((gains * disc[:len(gains)]).sum() / (ideal * disc[:len(ideal)]).sum())
        first = next((i for i, d in enumerate(docs[:mrr_k]) if judged.get(d, 0) > 0), None)
        mrr += 0.0 if first is None else 1.0 / (first + 1)
        rec += sum(1 for d in docs[:recall_k] if judged.get(d, 0) > 0) / len(rels)
    if n == 0:
        return {'queries_evaluated': 0}
    return {'nDCG@{}'.format(ndcg_k): round(ndcg / n, 5), 'MRR@{}'.format(mrr_k): round(mrr / n, 5),
            'R@{}'.format(recall_k): round(rec / n, 5), 'queries_evaluated': n}


def rcap_main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--qrel_file_path", type=str, required=True)
    parser.add_argument("--run_file_path", type=str, required=True)
    parser.add_argument("--cutoff", type=int, default=100, required=False)
    args = parser.parse_args(argv)
    qrels = read_qrels(args.qrel_file_path)
    results = {qid: dict(zip(docs, scores)) for qid, (docs, scores) in read_run(args.run_file_path).items()}
    print(recall_cap(qrels, results, [args.cutoff]))
