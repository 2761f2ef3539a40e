"""The closed-form plane / box cut of the GPU voxeliser (fibergen_amd/csrc/fg_plane_cut.h, compiled for the host) against
exact rational arithmetic: the inclusion-exclusion polynomial of the cut simplex evaluated with fractions.Fraction.
Covers normal components down to 1e-12 of the largest and exactly vanishing ones, coinciding kinks, and the definition
against a Monte-Carlo count."""
import ctypes
import os
import subprocess
from fractions import Fraction
from itertools import combinations

import numpy as np
import pytest

from helpers import emulation_build_flags

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "emu_cut.so")
    subprocess.check_call(["g++"] + emulation_build_flags() + ["-o", out, os.path.join(ROOT, "tests", "emulate", "emu_cut.cpp")])
    lib = ctypes.CDLL(out)
    lib.emu_cut_fraction.restype = ctypes.c_double
    lib.emu_cut_fraction.argtypes = [ctypes.c_double] * 4
    lib.emu_box_fraction.restype = ctypes.c_double
    lib.emu_box_fraction.argtypes = [dp, dp, dp]
    return lib


def exact_fraction(t, A):
    """volume fraction of {y in prod [0, A_i]-scaled unit box : sum y_i < t}: inclusion-exclusion, exact rationals"""
    t = Fraction(t)
    A = [Fraction(a) for a in A if a != 0]
    k = len(A)
    if k == 0:
        return 1.0 if t > 0 else 0.0
    p = lambda x: max(x, 0) ** k
    num = Fraction(0)
    for r in range(k + 1):
        for sub in combinations(A, r):
            num += (-1) ** r * p(t - sum(sub))
    den = Fraction(1)
    for a in A:
        den *= a
    fact = 1
    for i in range(2, k + 1):
        fact *= i
    return float(num / (fact * den))


def test_cut_against_exact_rationals(emu):
    rng = np.random.default_rng(1)
    worst = 0.0
    for trial in range(20000):
        A = np.sort(rng.random(3) * 10.0 ** rng.integers(-12, 1, 3))
        if rng.random() < 0.2:
            A[1] = A[2]
        if rng.random() < 0.1:
            A[0] = A[1]
        if rng.random() < 0.1:
            A[0] = 0.0
        if rng.random() < 0.03:
            A[1] = 0.0
            A[0] = 0.0
        t = rng.random() * A.sum()
        if rng.random() < 0.1:       # exactly on a kink
            t = float(rng.choice([A[0], A[1], A[2], A[1] + A[2], A[0] + A[2]]))
        worst = max(worst, abs(emu.emu_cut_fraction(t, *A) - exact_fraction(t, A)))
    assert worst < 1e-15


def test_box_fraction_definition(emu):
    """the fraction of the box on the side n . (y - xs) < 0, any signs of the normal, anisotropic box"""
    rng = np.random.default_rng(2)
    pts = rng.random((200000, 3))
    for trial in range(20):
        d = rng.random(3) * np.array([1.0, 2.0, 0.3]) + 0.05
        n = rng.standard_normal(3)
        if trial % 5 == 0:
            n[trial % 3] = 0.0
        n /= np.linalg.norm(n)
        xs = rng.random(3) * d
        got = emu.emu_box_fraction(xs.ctypes.data_as(dp), n.ctypes.data_as(dp), d.ctypes.data_as(dp))
        mc = float((((pts * d - xs) @ n) < 0).mean())
        assert abs(got - mc) < 5e-3
    # planes outside the box
    d = np.array([1.0, 1.0, 1.0])
    n = np.array([1.0, 0.0, 0.0])
    for x, want in ((-0.1, 0.0), (1.1, 1.0), (0.25, 0.25)):
        xs = np.array([x, 0.3, 0.3])
        assert emu.emu_box_fraction(xs.ctypes.data_as(dp), n.ctypes.data_as(dp), d.ctypes.data_as(dp)) == pytest.approx(want, abs=1e-16)
