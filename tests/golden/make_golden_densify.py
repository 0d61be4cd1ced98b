#!/usr/bin/env python
"""Golden vectors for `densify`: runs the REFERENCE function (tevatron/DHR/utils.py, loaded from /root/reference by
file path so that nothing else of the package is imported) on seeded inputs and stores inputs + outputs.
Run in the build container only:  python tests/golden/make_golden_densify.py"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_dhr_utils", "/root/reference/tevatron/DHR/utils.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(20260928)
out = {}
# case A: the production shape (BERT vocabulary 30522, 570 unused ids removed, 768 slices of 39), sparse non-negative
# "softmax x ReLU" like weights with exact ties (zeros) in most slices
a = np.zeros((3, 30522), np.float32)
for b in range(3):
    nz = rng.choice(30522, size=200, replace=False)
    a[b, nz] = rng.uniform(0.01, 3.0, size=200).astype(np.float32)
out["a_in"] = a
v, i = ref.densify(torch.from_numpy(a), 768)
out["a_val"], out["a_idx"] = v.numpy(), i.numpy()
# case B: small dims, dense signed values, fp16 input, ties between groups
b_in = rng.standard_normal((5, 3 + 8 * 5)).astype(np.float16)
b_in[0, 3 + 2] = b_in[0, 3 + 8 + 2] = np.float16(2.5)          # tie: first group must win
b_in[1, 3:] = np.float16(-1.0)                                   # all equal
out["b_in"] = b_in
v, i = ref.densify(torch.from_numpy(b_in), dims=8, remove_dims=3)
out["b_val"], out["b_idx"] = v.numpy(), i.numpy()
# case C: remove_dims = 0, one group (identity)
c_in = rng.standard_normal((2, 16)).astype(np.float32)
out["c_in"] = c_in
v, i = ref.densify(torch.from_numpy(c_in), dims=16, remove_dims=0)
out["c_val"], out["c_idx"] = v.numpy(), i.numpy()
# error behaviour
errs = []
for bad, kw in ((np.zeros((2, 3, 4), np.float32), dict(dims=4, remove_dims=0)), (np.zeros((2, 30), np.float32), dict(dims=7, remove_dims=1))):
    try:
        ref.densify(torch.from_numpy(bad), **kw)
        errs.append("")
    except ValueError as e:
        errs.append(str(e))
out["errors"] = np.array(errs)
np.savez_compressed(os.path.join(HERE, "densify_golden.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})
