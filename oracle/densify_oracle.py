"""ORACLE (test infrastructure, NOT product code): numpy restatement of the reference's `densify`
(/root/reference/tevatron/DHR/utils.py:5-22) and of the casts the encoder driver applies to its outputs
(/root/reference/tevatron/driver/encode.py:155-158, 180-183).  Only tests/ may import this module.

Parity pin: tests/golden/densify_golden.npz holds outputs of the reference function itself (torch, run in the
build container by tests/golden/make_golden_densify.py); tests/test_oracle_golden.py replays them.
"""
from __future__ import annotations

import numpy as np


def densify(lexical_reps: np.ndarray, dims: int = 768, strategy: str = "stride", remove_dims: int = 570):
    """utils.py:5-22.  [batch, vocab] -> (value [batch, dims], index [batch, dims] int64): column j of the output is
    the maximum over the vocabulary entries remove_dims + g*dims + j, g = 0 .. (vocab-remove_dims)/dims - 1, and the
    g that attains it (torch.max(1): the first maximal value)."""
    if lexical_reps.ndim != 2:                                                              # :11-12
        raise ValueError('Input lexical representation shape should be 2 (batch, vocab), but the input shape is {}'.format(lexical_reps.ndim))
    orig_dims = lexical_reps.shape[-1]
    if (orig_dims - remove_dims) % dims != 0:                                               # :14-16
        raise ValueError('Input lexical representation cannot be densified, please fix dims or remove_dims')
    batch = lexical_reps.shape[0]
    view = lexical_reps[:, remove_dims:].reshape(batch, -1, dims)                           # :20
    idx = view.argmax(1)                                                                    # :21 (argmax = first maximum)
    # the value is the element AT that index, as torch.max(1) returns it: numpy's own max(1) picks either zero of a (-0.0, +0.0) tie,
    # torch the first (checked against torch in the build container; found by tools/stress_modes.py in round 6)
    return np.take_along_axis(view, idx[:, None, :], 1)[:, 0, :], idx.astype(np.int64)


def densify_encoded(lexical_reps: np.ndarray, dims: int = 768, remove_dims: int = 570):
    """What lands in the index record (encode.py:155-158 / :180-183, :165-170): fp16 values, uint8 indices."""
    v, i = densify(lexical_reps, dims, "stride", remove_dims)
    return v.astype(np.float16), i.astype(np.uint8)
