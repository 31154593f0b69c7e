#!/usr/bin/env python
"""Compares the instruction streams of selected kernels between two hipcc -S outputs (labels and comments normalised).  A kernel
runs from its symbol to its .Lfunc_end label -- NOT to the first s_endpgm: kernels with an early-returning wave (the draw wave)
have several.  Usage: isa_diff.py base.s new.s 'base-substring=new-substring' ..."""
import difflib
import re
import sys


def kernels(path):
    out, name = {}, None
    for line in open(path):
        m = re.match(r'^(_ZN3cm3\w+):', line)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith('.Lfunc_end'):
            name = None
            continue
        if not t or t.startswith(('.', ';')):
            continue
        t = re.sub(r'\.LBB\d+_\d+', 'L', t)
        t = re.sub(r';.*', '', t).strip()
        if t:
            out[name].append(t)
    return out


def main():
    b, n = kernels(sys.argv[1]), kernels(sys.argv[2])
    for spec in sys.argv[3:]:
        bs, ns = spec.split('=')
        kb = [k for k in b if bs in k]
        kn = [k for k in n if ns in k]
        if len(kb) != 1 or len(kn) != 1:
            print("%-60s AMBIGUOUS / MISSING: base %s new %s" % (spec, kb, kn))
            continue
        same = b[kb[0]] == n[kn[0]]
        nd = sum(1 for l in difflib.unified_diff(b[kb[0]], n[kn[0]], lineterm='', n=0) if l[0] in '+-' and not l.startswith(('+++', '---')))
        st = [t for t in n[kn[0]] if t.startswith('global_store')]
        print("%-60s base %5d new %5d instr  identical %-5s diff-lines %5d  | new: %d stores, %d nt, %d s_endpgm"
              % (spec, len(b[kb[0]]), len(n[kn[0]]), same, nd, len(st), sum(t.endswith(' nt') for t in st), n[kn[0]].count('s_endpgm')))


if __name__ == "__main__":
    main()
