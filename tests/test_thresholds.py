"""cm3_amd/csrc/thresholds.h: the squared-distance thresholds that replace `sqrt(d2) < c` style tests in the particle
kernels are EXACT.  Derivation with rational arithmetic, then an exhaustive sweep over every non-negative float32
(2^31 values) for the three float tests, and a neighbourhood sweep for the float64 ones (monotonicity of the correctly
rounded square root covers the rest).

Reference expressions: multi-goal_spread.py:114-118 (is_collision: dist < 0.3), :125-129 (reached: -dist >= -0.05);
the skip distance is the build's exact early-out of core.py:180-196 (cm3_amd/csrc/particle.hip, contact_force): beyond it the
computed force is exactly +-0 -- 0.32 for the float32 soft-plus on the hardware exp2 / log2 units, 0.41 for the float32 libm
comparison build, 1.05 for float64; that the force IS zero there is the argument in particle.hip plus the GPU parity tests,
this file pins the equivalence of the squared test with the distance test."""
import os
import re
from fractions import Fraction

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "cm3_amd", "csrc", "thresholds.h")


def header_constants():
    out = {}
    pat = re.compile(r"=\s*(\S+?)f?;\s*//\s*CM3_THRESH (f32|f64) (\w+) (\S+)")
    for line in open(HEADER):
        m = pat.search(line)
        if m:
            out[(m.group(2), m.group(3))] = (float.fromhex(m.group(1)), float(m.group(4)))
    return out


def t_lt(c, dtype):
    """min{x of dtype : RN(sqrt(x)) >= c}: RN(s) >= c  <=>  s >= midpoint(prev(c), c) (an exact tie s == midpoint would
    need the midpoint's square to be a dtype number; checked by the sweeps)."""
    c = dtype(c)
    prev = np.nextafter(c, dtype(0))
    m2 = ((Fraction(float(prev)) + Fraction(float(c))) / 2) ** 2
    x = dtype(float(m2))
    while Fraction(float(x)) < m2:
        x = np.nextafter(x, dtype(np.inf))
    while Fraction(float(np.nextafter(x, dtype(0)))) >= m2:
        x = np.nextafter(x, dtype(0))
    return x


def constants_of(dtype):
    tag = "f32" if dtype is np.float32 else "f64"
    consts = header_constants()
    coll_c = dtype(0.15) + dtype(0.15)                  # kDistMin as the kernels and the reference form it
    skip_c = dtype(consts[(tag, "skip")][1])
    reach_c = dtype(0.05)
    return consts, tag, coll_c, skip_c, reach_c


def test_header_has_all_constants():
    """three per precision, plus the float32 skip distance of the -DCM3_F32_LIBM_SOFTPLUS comparison build"""
    consts = header_constants()
    assert sorted(consts) == sorted([(t, k) for t in ("f32", "f64") for k in ("coll", "skip", "reach")] + [("f32", "skip_libm")])
    assert consts[("f32", "skip")][1] == 0.32 and consts[("f32", "skip_libm")][1] == 0.41 and consts[("f64", "skip")][1] == 1.05


def test_constants_equal_the_exact_derivation():
    for dtype in (np.float32, np.float64):
        consts, tag, coll_c, skip_c, reach_c = constants_of(dtype)
        assert dtype(consts[(tag, "coll")][0]) == t_lt(coll_c, dtype)
        assert dtype(consts[(tag, "skip")][0]) == t_lt(skip_c, dtype)
        assert dtype(consts[(tag, "reach")][0]) == t_lt(np.nextafter(reach_c, dtype(1)), dtype)
        if tag == "f32":
            assert dtype(consts[(tag, "skip_libm")][0]) == t_lt(dtype(consts[(tag, "skip_libm")][1]), dtype)
        # the header's values are exactly representable in the working precision
        for k in ("coll", "skip", "reach"):
            assert float(dtype(consts[(tag, k)][0])) == consts[(tag, k)][0]


def _check(x, dtype, consts, tag, coll_c, skip_c, reach_c):
    s = np.sqrt(x)
    bad = np.count_nonzero((s < coll_c) != (x < dtype(consts[(tag, "coll")][0])))
    bad += np.count_nonzero((~(s >= skip_c)) != (~(x >= dtype(consts[(tag, "skip")][0]))))
    if (tag, "skip_libm") in consts:
        c2, c = consts[(tag, "skip_libm")]
        bad += np.count_nonzero((~(s >= dtype(c))) != (~(x >= dtype(c2))))
    bad += np.count_nonzero(((dtype(0) - s) >= dtype(-0.05)) != (x < dtype(consts[(tag, "reach")][0])))
    return int(bad)


def test_float32_exhaustive():
    """every non-negative float32 incl. +inf, plus the NaNs of one chunk"""
    consts, tag, coll_c, skip_c, reach_c = constants_of(np.float32)
    chunk = 1 << 24
    bad = 0
    for start in range(0, 0x7f800001, chunk):
        bits = np.arange(start, min(start + chunk, 0x7f800001), dtype=np.uint32)
        bad += _check(bits.view(np.float32), np.float32, consts, tag, coll_c, skip_c, reach_c)
    nan = np.arange(0x7f800001, 0x7f800001 + 4096, dtype=np.uint32).view(np.float32)
    with np.errstate(invalid="ignore"):
        bad += _check(nan, np.float32, consts, tag, coll_c, skip_c, reach_c)
    assert bad == 0


def test_float64_neighbourhoods():
    consts, tag, coll_c, skip_c, reach_c = constants_of(np.float64)
    for k in ("coll", "skip", "reach"):
        t = np.float64(consts[(tag, k)][0])
        base = int(np.array([t]).view(np.uint64)[0])
        bits = np.arange(base - (1 << 20), base + (1 << 20), dtype=np.uint64)
        assert _check(bits.view(np.float64), np.float64, consts, tag, coll_c, skip_c, reach_c) == 0
    special = np.array([0.0, 5e-324, 1e-300, 1.0, 1e300, np.inf, np.nan])
    with np.errstate(invalid="ignore"):
        assert _check(special, np.float64, consts, tag, coll_c, skip_c, reach_c) == 0
