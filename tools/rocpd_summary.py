#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, share) of a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats` in ROCm 7.2 writes <name>_results.db).  Writes a small text table
that is committed under profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, (end - start) from kernels").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values()) or 1
    lines = ["# kernel-trace summary of %s (durations in microseconds)" % db,
             "%-88s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct")]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%-88s %8d %14.1f %12.2f %12.2f %12.2f %6.2f%%" % (name[:88], a[0], a[1] / 1e3, a[1] / a[0] / 1e3,
                                                                      a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
