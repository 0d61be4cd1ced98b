"""Host TREC helpers (dhr_amd/retrieval/trec.py): shard-file merge and capped recall.

The merge is checked against the shard-reduce restatement in oracle/gip_oracle.py and by hand-built
cases; the capped recall against values worked out by hand from the definition
(reference retrieval/evaluation/custom_metrics.py:46-55)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from dhr_amd.retrieval import trec

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_shards(tmp_path, n_shards, n_q, per_shard, seed):
    rng = np.random.default_rng(seed)
    truth = {}
    for s in range(n_shards):
        run = {}
        for q in range(n_q):
            scores = np.sort(rng.random(per_shard) * 50)[::-1]
            docs = ['d{}_{}'.format(s, i) for i in rng.permutation(per_shard)]
            run['q{}'.format(q)] = (docs, [float(x) for x in scores])
            truth.setdefault('q{}'.format(q), []).extend(zip(docs, scores))
        trec.write_run(str(tmp_path / 'result{:02d}.trec'.format(s)), run, 'shard')
    return truth


def test_merge_cli_matches_global_sort(tmp_path):
    truth = _write_shards(tmp_path, 3, 5, 20, seed=1)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'retrieval', 'merge.result.py'), '--total_shrad', '3',
                          '--topk', '25', '--run_name', 'x'], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    merged = trec.read_run(str(tmp_path / 'result.trec'))
    assert list(merged) == ['q{}'.format(q) for q in range(5)]
    for qid, pairs in truth.items():
        want = sorted(pairs, key=lambda p: -p[1])[:25]
        docs, scores = merged[qid]
        assert docs == [d for d, _ in want]
        assert scores == [float(s) for _, s in want]
    lines = open(tmp_path / 'result.trec').read().splitlines()
    assert lines[0].split(' ')[1] == 'Q0' and lines[0].split(' ')[3] == '1' and lines[0].endswith(' x')
    assert [int(l.split(' ')[3]) for l in lines[:25]] == list(range(1, 26))


def test_merge_matches_oracle_shard_reduce(tmp_path):
    """Same text as the restated reference merge when no two scores tie."""
    from oracle import gip_oracle
    _write_shards(tmp_path, 4, 6, 15, seed=3)
    texts = [open(tmp_path / 'result{:02d}.trec'.format(s)).read() for s in range(4)]
    want = gip_oracle.merge_results(texts, topk=20, run_name='dhr')
    trec.merge_main(['--total_shrad', '4', '--topk', '20', '--dir', str(tmp_path)])
    assert open(tmp_path / 'result.trec').read() == want


def test_merge_ties_keep_shard_order_and_ragged_shards():
    runs = [{'a': (['x', 'y'], [2.0, 1.0])}, {'a': (['z'], [2.0]), 'b': (['w'], [0.5])}, {}]
    merged = trec.merge_runs(runs, 10)
    assert merged['a'] == (['x', 'z', 'y'], [2.0, 2.0, 1.0])
    assert merged['b'] == (['w'], [0.5])
    assert trec.merge_runs(runs, 1)['a'] == (['x'], [2.0])


def test_read_run_rejects_malformed_line(tmp_path):
    p = tmp_path / 'bad.trec'
    p.write_text('q Q0 d 1 0.5\n')
    with pytest.raises(ValueError):
        trec.read_run(str(p))


def test_recall_cap_by_hand():
    qrels = {'q1': {'a': 1, 'b': 1, 'c': 0, 'd': 2}, 'q2': {'e': 1}, 'q3': {'f': 1}}
    results = {'q1': {'a': 9.0, 'c': 8.0, 'x': 7.0, 'd': 6.0, 'b': 1.0},
               'q2': {'y': 3.0, 'e': 2.0}}
    # q1: 3 relevant; top-2 = a,c -> 1/min(3,2); top-4 = a,c,x,d -> 2/3.  q2: top-2 -> 1/1; top-1 -> 0.
    # Sum over run queries, divided by the number of qrel queries (3).
    got = trec.recall_cap(qrels, results, [1, 2, 4])
    assert got == {'R_cap@1': round((1 / 1 + 0) / 3, 5), 'R_cap@2': round((0.5 + 1.0) / 3, 5),
                   'R_cap@4': round((2 / 3 + 1.0) / 3, 5)}


def test_rcap_cli(tmp_path):
    (tmp_path / 'qrels.tsv').write_text('q1\t0\ta\t1\nq1\t0\tb\t1\nq2\t0\tc\t1\n')
    trec.write_run(str(tmp_path / 'run.trec'), {'q1': (['a', 'z'], [2.0, 1.0]), 'q2': (['z', 'c'], [2.0, 1.0])}, 'r')
    out = subprocess.run([sys.executable, '-m', 'retrieval.rcap_eval', '--qrel_file_path', str(tmp_path / 'qrels.tsv'),
                          '--run_file_path', str(tmp_path / 'run.trec'), '--cutoff', '2'],
                         cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == str({'R_cap@2': 0.75})


def test_effectiveness_hand_worked(tmp_path):
    """nDCG@10 / MRR@10 / R@1000 (trec.effectiveness: the evaluator behind bench.py --qrels) against values worked out by hand from
    trec_eval's definitions."""
    qrels_path = tmp_path / 'qrels.tsv'
    qrels_path.write_text('q1 0 a 1\nq1 0 b 0\nq1 0 c 2\nq2\t0\tx\t1\nq3 0 z 0\n')
    qrels = trec.read_qrels_any(str(qrels_path))
    assert qrels == {'q1': {'a': 1, 'b': 0, 'c': 2}, 'q2': {'x': 1}, 'q3': {'z': 0}}
    run = {'q1': (['b', 'c', 'u', 'a'], [4.0, 3.0, 2.0, 1.0]),       # gains 0, 2, 0, 1
           'q2': (['y', 'w'], [2.0, 1.0]),                           # the relevant doc is not retrieved
           'q3': (['z'], [1.0]),                                     # no relevant judgment: skipped, as trec_eval does
           'q4': (['z'], [1.0])}                                     # not judged at all: skipped
    e = trec.effectiveness(qrels, run)
    dcg = 2 / np.log2(3) + 1 / np.log2(5)
    idcg = 2 / np.log2(2) + 1 / np.log2(3)
    assert e['queries_evaluated'] == 2
    assert e['nDCG@10'] == round((dcg / idcg + 0.0) / 2, 5)
    assert e['MRR@10'] == round((1 / 2 + 0.0) / 2, 5)
    assert e['R@1000'] == round((2 / 2 + 0 / 1) / 2, 5)
    assert trec.effectiveness(qrels, {'q9': (['a'], [1.0])}) == {'queries_evaluated': 0}
