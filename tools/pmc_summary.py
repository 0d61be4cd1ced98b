#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc counter CSVs under a directory per (run, kernel, counter) for the bound-GEMM kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for run in sorted(os.listdir(root)):
    d = os.path.join(root, run)
    if not os.path.isdir(d):
        continue
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row.get("Kernel_Name", "")
                if "gemm_filter" not in k:
                    continue
                name = "g8p" if "_g8p_" in k else "g8" if "_g8_" in k else "wx" if "_wx_" in k else ("sparse" if "sparse" in k else "v3")
                a = acc[(name, row["Counter_Name"])]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    for (name, c), (v, n) in sorted(acc.items()):
        print(f"{run:10s} {name:7s} {c:28s} per-launch {v / max(n, 1):16.1f}  ({n} launches)")
