#!/usr/bin/env python
"""Per-kernel mean of every PMC counter found under a tools/pmc_run.sh output directory."""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "k_particle_step"
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            name = r.get("Kernel_Name", "")
            if want not in name:
                continue
            short = name.split("(")[0][-60:]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print("kernel:", k)
        for c, v in sorted(cs.items()):
            print("  %-24s n=%6d mean=%16.2f min=%14.1f max=%14.1f" % (c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
