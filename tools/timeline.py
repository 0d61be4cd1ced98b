#!/usr/bin/env python
"""Kernel timeline of the tail of a rocprofv3 --kernel-trace database (rocpd sqlite): start offset, duration and the idle gap in front of
every kernel, for reading where a short multi-kernel step spends its time.  usage: timeline.py <results.db> [last_n_kernels]"""
import sqlite3
import sys


def main(db, last=150):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[-last:]
    t0 = rows[0][1]
    prev_end = t0
    busy = 0
    for name, s, e in rows:
        print("%9.1f us  +%8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:70]))
        busy += e - s
        prev_end = max(prev_end, e)
    print("span %.1f us, kernels %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 150)
