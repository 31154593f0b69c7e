"""Per-kernel register / LDS / scratch figures out of `hipcc -Rpass-analysis=kernel-resource-usage` stderr.
usage: resource_usage.py <file> [substring of the mangled kernel name ...]"""
import re
import sys


def parse(path):
    out, cur = [], None
    for line in open(path):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            out.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:\[]*?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return out


if __name__ == "__main__":
    keys = sys.argv[2:]
    for k in parse(sys.argv[1]):
        if keys and not any(s in k["name"] for s in keys):
            continue
        print("%-64s VGPRs %3s AGPRs %3s occ %s scratch %3s sgpr-spill %3s vgpr-spill %3s LDS %6s" % (
            k["name"][-64:], k.get("VGPRs"), k.get("AGPRs"), k.get("Occupancy"), k.get("ScratchSize"), k.get("SGPRs Spill"),
            k.get("VGPRs Spill"), k.get("LDS Size")))
