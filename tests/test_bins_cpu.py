"""The arithmetic of the ICP loop's bins (libwave_amd/csrc/wm_bins.hpp) on the CPU, through the C ABI's host-only
wm_debug_bins_sum: a double is cut into three signed 40-bit limbs of trunc(x * 2^56), limbs are added as integers, the
totals are turned back into a double.  Claims checked: the limb totals ARE the exact sum of the truncated addends
(against Python's exact rational arithmetic), whatever the order of addition; the double that comes back is the
correctly rounded exact sum for addends that fit the format exactly (|x| >= 2^-4: every per-wave sum of coordinates and
their products that matters), and within 2^-56 per addend otherwise.  (pcl::IterativeClosestPoint's per-iteration sums,
wave_matching/src/icp.cpp:95,116,126 -> the device path's k_bins_solve.)"""
import ctypes as C
import math
from fractions import Fraction

import numpy as np
import pytest

from libwave_amd import capi as wm


def _bins_sum(x, perm=None):
    L = wm.lib()
    L.wm_debug_bins_sum.argtypes = [C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_uint), C.POINTER(C.c_double),
                                    C.POINTER(C.c_longlong)]
    L.wm_debug_bins_sum.restype = C.c_int
    x = np.ascontiguousarray(x, np.float64)
    out = C.c_double()
    limbs = (C.c_longlong * 3)()
    p = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, np.uint32)
        p = perm.ctypes.data_as(C.POINTER(C.c_uint))
    rc = L.wm_debug_bins_sum(x.ctypes.data_as(C.POINTER(C.c_double)), len(x), p, C.byref(out), limbs)
    return rc, out.value, [int(v) for v in limbs]


def _exact_truncated_sum(x):
    """sum of trunc(x * 2^56) as exact integers (what the limbs must add up to), in units of 2^-56"""
    tot = 0
    for v in x:
        f = Fraction(float(v)) * (1 << 56)
        tot += int(f) if f >= 0 else -int(-f)      # trunc toward zero
    return tot


def test_limbs_are_the_exact_sum_in_any_order():
    rng = np.random.default_rng(7)
    # per-wave sums as the search kernel forms them: counts, coordinates of a 100 m scene, their products, d^2 sums,
    # of both signs, over nine orders of magnitude
    x = np.concatenate([rng.integers(0, 65, 500).astype(np.float64), rng.normal(0, 3000, 4000),
                        rng.normal(0, 2.0e5, 4000), rng.uniform(0, 40, 2000), rng.normal(0, 1e-3, 500)])
    want = _exact_truncated_sum(x)
    rc, got, limbs = _bins_sum(x)
    assert rc == 0
    assert limbs[0] + (limbs[1] << 40) + (limbs[2] << 80) == want
    for seed in range(3):
        perm = np.random.default_rng(seed).permutation(len(x))
        rc2, got2, limbs2 = _bins_sum(x, perm)
        assert rc2 == 0 and limbs2[0] + (limbs2[1] << 40) + (limbs2[2] << 80) == want
        assert got2 == got                                 # the same double, bit for bit, whatever the order


def test_the_double_is_the_correctly_rounded_exact_sum():
    rng = np.random.default_rng(11)
    # addends with no bits below 2^-56 (|x| >= 2^-4 and 53-bit mantissas): nothing is truncated
    x = rng.normal(0, 1.0e4, 20000)
    x = x[np.abs(x) >= 0.0625]
    rc, got, _ = _bins_sum(x)
    assert rc == 0
    exact = sum(Fraction(float(v)) for v in x)
    assert got == float(exact)                             # Fraction -> float rounds to nearest: the same double
    assert got == math.fsum(x)
    # small addends lose what lies below 2^-56, never more
    y = rng.normal(0, 1e-6, 5000)
    rc, got, _ = _bins_sum(y)
    assert rc == 0 and abs(got - math.fsum(y)) <= len(y) * 2.0 ** -56


def test_cancellation_and_extremes():
    big = 3.0e18                                           # just inside 2^62
    rc, got, _ = _bins_sum([big, 1.0, -big, 0.25, -1.0])
    assert rc == 0 and got == 0.25
    rc, got, _ = _bins_sum([-7.5] * 1000 + [7.5] * 999)
    assert rc == 0 and got == -7.5
    assert _bins_sum([])[0] == 0 and _bins_sum([])[1] == 0.0
    for bad in (float("nan"), float("inf"), 5.0e18):       # the device poisons the bin for these
        assert _bins_sum([1.0, bad])[0] == wm.WM_ERR_ARG
