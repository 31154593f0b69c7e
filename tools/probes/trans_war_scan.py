"""Scan gfx950 assembly for a write-after-read on the source of a transcendental instruction: `v_log_f32 vD, vS` followed within
`--window` instructions by a VALU instruction whose destination is vS.  usage: trans_war_scan.py file.s [--window 2]"""
import re
import sys

TRANS = re.compile(r"^\s*(v_(?:exp|log|rcp|rsq|sqrt|sin|cos)(?:_legacy|_iflag)?_(?:f16|f32|f64|bf16))(?:_e32|_e64)?\s+(.*)$")
INSTR = re.compile(r"^\s*([a-z][a-z0-9_]+)\s*(.*)$")


def regs(tok):
    """v80 -> {80}; v[80:81] -> {80, 81}; |v81| / -v81 -> {81}"""
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def scan(path, window):
    kernel, hits, lines = None, [], open(path).read().splitlines()
    body = []
    for n, line in enumerate(lines):
        if re.match(r"^[_A-Za-z][\w$.]*:\s*(;.*)?$", line) and not line.startswith(".L"):
            kernel = line.split(":")[0]
        code = line.split(";")[0].rstrip()
        if not code.strip() or code.strip().startswith(".") or code.strip().endswith(":"):
            continue
        body.append((n + 1, kernel, code))
    for i, (n, k, code) in enumerate(body):
        m = TRANS.match(code)
        if not m:
            continue
        ops = m.group(2).split(",")
        src = set()
        for o in ops[1:]:
            src |= regs(o)
        seen = 0
        for n2, k2, c2 in body[i + 1:]:
            mi = INSTR.match(c2)
            if not mi:
                continue
            name = mi.group(1)
            if name.startswith("s_nop"):
                seen += int(mi.group(2) or 0) + 1
            else:
                seen += 1
            if seen > window:
                break
            if name.startswith("v_") and not name.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                dst = regs(mi.group(2).split(",")[0])
                if dst & src:
                    hits.append((k, n, code.strip(), n2, c2.strip(), seen))
                    break
            if name.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                break
    return hits


if __name__ == "__main__":
    window = 2
    args = sys.argv[1:]
    if "--window" in args:
        i = args.index("--window")
        window = int(args[i + 1])
        del args[i:i + 2]
    for path in args:
        hits = scan(path, window)
        by = {}
        for h in hits:
            by.setdefault(h[0], []).append(h)
        print("%s: %d trans-source overwrites within %d wait states, in %d kernels" % (path, len(hits), window, len(by)))
        for k, hs in by.items():
            print("  %s: %d" % (k, len(hs)))
            for h in hs[:6]:
                print("      line %d  %-44s -> line %d  %s   (distance %d)" % (h[1], h[2], h[3], h[4], h[5]))
