#!/usr/bin/env python
"""ISA lint of the built library's device code (round 5).

Finding (profiles/r05_policy_fault.txt): on MI355X a `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]` -- a packed float32 instruction whose
LOW result takes the HIGH dword of a source -- returned +-0 as its low result in lanes 48..63, a handful of times per launch, while
the SIMD's other wave ran a float16 matrix layer; wait states around it did not help, the same product as two v_mul_f32 or as a
packed multiply WITHOUT a cross-half select was clean.  The compiler's SLP vectoriser creates that form out of scalar code (a pair
of unrelated multiplies packed, then one element of the pair broadcast).  csrc/build.sh therefore builds the physics translation
units with -fno-slp-vectorize, and this lint fails if any packed float32 instruction with a low-half cross select (a 1 in
`op_sel:[...]`) is left in a device code object.

    python tools/isa_lint.py cm3_amd/csrc/_obj/*.o         (exit status 1 and the offending lines if any)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
BAD = re.compile(r"\bv_pk_(?:mul|add|fma|mov)_(?:f32|b32)\b.*\bop_sel:\[[01,]*1[01,]*\]")


def device_isa(obj):
    with tempfile.TemporaryDirectory() as d:
        co, fb = os.path.join(d, "dev.co"), os.path.join(d, "dev.hipfb")
        # the host object carries the device binary as a clang offload bundle in its .hip_fatbin section
        r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, obj, os.path.join(d, "copy.o")],
                           capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(fb):
            return None
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "-input=" + fb, "-output=" + co, "-unbundle"], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            return None
        r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True)
        return r.stdout if r.returncode == 0 else None


def lint(objs):
    bad, seen = [], 0
    for obj in objs:
        isa = device_isa(obj)
        if isa is None:
            continue
        seen += 1
        kernel = "?"
        for line in isa.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                kernel = m.group(1)
            elif BAD.search(line):
                bad.append((os.path.basename(obj), kernel, line.strip()))
    return seen, bad


if __name__ == "__main__":
    seen, bad = lint(sys.argv[1:])
    print("isa_lint: %d device code objects, %d packed float32 instructions with a low-half cross select" % (seen, len(bad)))
    for o, k, l in bad[:40]:
        print("  %s  %s\n      %s" % (o, k[:90], l))
    sys.exit(1 if bad or not seen else 0)
