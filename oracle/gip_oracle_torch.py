"""ORACLE (test infrastructure, NOT product code): the reference's per-query loop restated with the reference's OWN ops
(torch on the CPU), for the timed `cpu_baseline` legs of bench.py.

/root/reference/retrieval/gip_retrieval.py:115-126 (one query at a time: mask = c_idx == q_idx; tmp = mask * c_val;
einsum('ij,j->i'); topk) and :70-79 (dense-only: einsum + argsort).  The thread count is the reference's:
torch.set_num_threads(1) for --batch 1, all cores otherwise (:255-259).  Checked against oracle/gip_oracle.py (numpy, pinned to
the reference's outputs) in tests/test_oracle_golden.py::test_torch_restatement_equals_numpy_oracle.
"""
from __future__ import annotations

import time

import numpy as np

LAST_THREADS = 0          # torch.get_num_threads() as read INSIDE the last gip_loop (the setting is restored on the way out)


def gip_loop(q32: np.ndarray, qi, c32: np.ndarray, ci, k: int, threads: int):
    """-> (seconds per query, rows [Q, k]).  q32 / c32 fp32 (the reference's CPU dtype), qi / ci unpadded index arrays or None."""
    import torch
    old = torch.get_num_threads()
    torch.set_num_threads(max(1, threads))
    global LAST_THREADS
    LAST_THREADS = torch.get_num_threads()
    try:
        c = torch.from_numpy(c32)
        q = torch.from_numpy(q32)
        if ci is not None:
            cls_dim = c.shape[1] - ci.shape[1]
            cip = torch.nn.functional.pad(torch.from_numpy(ci), (0, cls_dim), value=1)      # :110-113
            qip = torch.nn.functional.pad(torch.from_numpy(qi), (0, cls_dim), value=1)
        rows = []
        t0 = time.perf_counter()
        for i in range(q.shape[0]):
            if ci is not None:
                mask = cip == qip[i]                                                         # :119
                tmp = mask * c
                s = torch.einsum("ij,j->i", tmp, q[i])                                       # :120
                rows.append(torch.topk(s, min(k, s.shape[0])).indices.numpy())              # :123
            else:
                s = torch.einsum("ij,j->i", c, q[i])                                         # :74
                rows.append(torch.argsort(s, descending=True)[:k].numpy())                   # :75
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(old)
    return dt / q.shape[0], np.stack(rows)
