"""Drop-in for castorini/dhr's retrieval/index.py: merge `<prefix>.split*.pt` into `<prefix>.index.pt`.

Same flags and the same output pickle ([value, index|0, docids], protocol 4) as
/root/reference/retrieval/index.py:18-47.  One deliberate difference: the split files are merged in
SORTED name order -- the reference iterates `glob.glob` order, which is filesystem dependent (the
survey probe saw split00, split02, split01), so its row order is not reproducible."""
from __future__ import annotations

import argparse
import glob
import os
import pickle

import numpy as np


def merge_splits(index_path: str, index_prefix: str):
    corpus_files = sorted(glob.glob(os.path.join(index_path, index_prefix + '.split*.pt')))
    if not corpus_files:
        raise FileNotFoundError(f"no {index_prefix}.split*.pt under {index_path}")
    corpus_embs, corpus_arg_idxs, docids = [], [], []
    for corpus_file in corpus_files:
        with open(corpus_file, 'rb') as f:
            print('Load index: {}...'.format(corpus_file))
            corpus_emb, corpus_arg_idx, docid = pickle.load(f)
        corpus_embs.append(corpus_emb)
        corpus_arg_idxs.append(corpus_arg_idx)
        docids += docid
    print('Merge index ...')
    if any(a is None or np.isscalar(a) for a in corpus_arg_idxs):
        corpus_arg_idxs = 0                      # dense models carry no index array (index.py:40-43)
    else:
        corpus_arg_idxs = np.concatenate(corpus_arg_idxs, axis=0)
    return [np.concatenate(corpus_embs, axis=0), corpus_arg_idxs, docids]


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--index_prefix", type=str, default='msmarco-passage')
    parser.add_argument("--emb_dim", type=int, default=768)
    parser.add_argument("--index_path", type=str, required=True)
    args = parser.parse_args(argv)
    merged = merge_splits(args.index_path, args.index_prefix)
    with open(os.path.join(args.index_path, args.index_prefix + '.index.pt'), 'wb') as f:
        pickle.dump(merged, f, protocol=4)


if __name__ == "__main__":
    main()
