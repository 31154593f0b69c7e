#!/usr/bin/env python
"""Per-kernel mean of every PMC counter found under a tools/pmc_run.sh output directory."""
import collections
import csv
import glob
import os
import sys


def traffic_json(acc, tag, out_path):
    """Appends {tag: {...}} to profiles/pmc_traffic.json: per-launch HBM bytes as the MI355X guide prescribes --
    (FETCH_SIZE [x2 on gfx950: it reports half of a wide coalesced read stream] + WRITE_SIZE) * 1024."""
    import json
    data = {}
    if os.path.exists(out_path):
        data = json.load(open(out_path))
    for k, cs in acc.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            f = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"])
            w = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"])
            data[tag] = {"kernel": k.strip(), "launches": len(cs["FETCH_SIZE"]), "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w,
                         "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
                         "hbm_bytes_per_launch_uncorrected": (f + w) * 1024.0,
                         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_run.sh); "
                                 "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B reads as 64 B)"}
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)


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
    if len(sys.argv) > 4:
        traffic_json(acc, sys.argv[3], sys.argv[4])
    for k, cs in acc.items():
        print("kernel:", k)
        for c, v in sorted(cs.items()):
            print("  %-24s n=%6d mean=%16.2f min=%14.1f max=%14.1f" % (c, len(v), sum(v) / len(v), min(v), max(v)))


if __name__ == "__main__":
    main()
