#!/usr/bin/env python
"""Prints one kernel's instruction stream from a `hipcc -S --cuda-device-only` output (comments and directives stripped, branch
labels kept), optionally only the lines matching given prefixes.  Usage: isa_dump.py file.s <symbol-substring> [prefix ...]"""
import re
import sys


def kernel_body(path, sub):
    body, on = [], False
    for line in open(path):
        if not on:
            m = re.match(r'^(_ZN3cm3\w+):', line)
            if m and sub in m.group(1):
                on = True
            continue
        t = line.strip()
        if t.startswith('.Lfunc_end'):
            break
        if t.startswith('.LBB'):
            body.append(t)
            continue
        if not t or t.startswith(('.', ';')):
            continue
        body.append(re.sub(r';.*', '', t).strip())
    return body


if __name__ == "__main__":
    b = kernel_body(sys.argv[1], sys.argv[2])
    pre = tuple(sys.argv[3:])
    for k, t in enumerate(b):
        if not pre or t.startswith(pre):
            print(k, t)
    print("# %d lines" % len(b))
