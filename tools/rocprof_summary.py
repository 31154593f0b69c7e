#!/usr/bin/env python
"""Summarises a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db or *_kernel_stats.csv) as text:
per-kernel calls / total / average / min / max / share, the table committed under profiles/."""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    return [(n, int(k), float(t), float(a), float(lo), float(hi)) for n, k, t, a, lo, hi in rows]


def from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]),
                    float(r["MinNs"]), float(r["MaxNs"])))
    return out


def main():
    src = sys.argv[1]
    if os.path.isdir(src):
        dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
        csvs = glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True)
        src = (csvs or dbs)[0]
    rows = from_csv(src) if src.endswith(".csv") else from_db(src)
    total = sum(r[2] for r in rows) or 1.0
    print("# rocprofv3 --kernel-trace --stats summary of %s" % os.path.basename(src))
    print("%-72s %8s %14s %12s %10s %10s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "share"))
    for n, k, t, a, lo, hi in rows:
        short = n if len(n) <= 72 else n[:69] + "..."
        print("%-72s %8d %14.0f %12.1f %10.0f %10.0f %6.2f%%" % (short, k, t, a, lo, hi, 100.0 * t / total))


if __name__ == "__main__":
    main()
