"""The voxeliser's CHECKER (oracle/c/fg_voxel_ref.cpp: the reference's recursive integratePhiVoxel F:16622-16752 with
halfspace_box_cut_volume F:1385-1577, restated on the host) pinned on known answers, and the question which phase
fractions the reference's current code produces for its own Hashin demo."""
import math

import numpy as np
import pytest

from oracle import voxel_oracle
from oracle.ls_oracle import LSOracle


class Fiber:
    def __init__(self, kind, c, a, L, R, material):
        self.kind, self.c, self.a, self.L, self.R, self.material = kind, c, a, L, R, material


def test_checker_known_answers():
    sph = [Fiber("capsule", [.5, .5, .5], [1, 0, 0], 0.0, 0.3, 1)]
    phi, nrm, real = voxel_oracle.voxelize(sph, (32, 32, 32), (1, 1, 1), (0, 0, 0), 2, 0, want_normals=True)
    exact = 4 / 3 * math.pi * 0.3 ** 3
    assert phi[0].min() == 1.0                      # matrix: all ones before normalisation
    assert abs(phi[1].mean() - exact) / exact < 5e-4
    assert real[1] == pytest.approx(exact, rel=1e-14)
    assert ((phi[1] >= 0) & (phi[1] <= 1)).all()
    # normals: unit, pointing out of the inclusion (F:5286-5294)
    assert np.allclose((nrm * nrm).sum(axis=0), 1.0)
    assert nrm[0, 31, 16, 16] > 0.99 and nrm[0, 0, 16, 16] < -0.99
    assert np.allclose(phi[1], phi[1][::-1]) and np.allclose(phi[1], phi[1].transpose(1, 0, 2))   # cube group
    # capsule: total length L, cylinder part L - 4/3 R (F:5256-5258): volume = pi R^2 L
    cap = [Fiber("capsule", [.5, .5, .5], [0, 0, 1], 0.6, 0.2, 1)]
    phi, _, real = voxel_oracle.voxelize(cap, (32, 32, 32), (1, 1, 1), (0, 0, 0), 2, 0)
    assert real[1] == pytest.approx(math.pi * 0.2 ** 2 * 0.6, rel=1e-13)
    assert abs(phi[1].mean() - real[1]) / real[1] < 2e-3
    # the three half spaces of demo/elasticity/laminate/project.xml:31-36
    fib = [Fiber("halfspace", [0.0, .5, .5], [1, 0, 0], 0, 0.25, 0),
           Fiber("halfspace", [0.2, .5, .5], [-1, 0, 0], 0, 0.25, 1),
           Fiber("halfspace", [0.5, .5, .5], [-1, 0, 0], 0, 0.25, 2)]
    phi, _, _ = voxel_oracle.voxelize(fib, (10, 1, 1), (1, 1, 1), (0, 0, 0), 3, 0)
    out = voxel_oracle.normalize_phi(phi)[:, :, 0, 0]
    assert np.allclose(out[0], [1, 1, 0, 0, 0, 0, 0, 0, 0, 0], atol=1e-15)
    assert np.allclose(out[1], [0, 0, 1, 1, 1, 0, 0, 0, 0, 0], atol=1e-15)
    assert np.allclose(out[2], [0, 0, 0, 0, 0, 1, 1, 1, 1, 1], atol=1e-15)
    # a plane cutting voxels obliquely: exact volume fraction of the half space x+y < 1 is 1/2
    obl = [Fiber("halfspace", [.5, .5, .5], [1, 1, 0], 0, 0.25, 1)]
    phi, _, _ = voxel_oracle.voxelize(obl, (8, 8, 2), (1, 1, 1), (0, 0, 0), 2, 0)
    assert phi[1].mean() == pytest.approx(0.5, abs=1e-14)
    with pytest.raises(RuntimeError, match="zero normal"):
        voxel_oracle.voxelize([Fiber("halfspace", [0, 0, 0], [0, 0, 0], 0, 0.1, 1)], (4, 4, 4), (1, 1, 1), (0, 0, 0), 2, 0)


def _supersampled_sphere(n, R, sub=16):
    """volume fractions of a centred sphere by brute force: sub^3 sample points in every voxel the surface can touch --
    independent of the plane-cut algorithm"""
    x = (np.arange(n) + 0.5) / n - 0.5
    r = np.sqrt(x[:, None, None] ** 2 + x[None, :, None] ** 2 + x[None, None, :] ** 2)
    phi = (r < R).astype(np.float64)
    band = np.argwhere(np.abs(r - R) < 0.5 * math.sqrt(3.0) / n)
    s = ((np.arange(sub) + 0.5) / sub - 0.5) / n
    for i, j, k in band:
        d2 = (x[i] + s)[:, None, None] ** 2 + (x[j] + s)[None, :, None] ** 2 + (x[k] + s)[None, None, :] ** 2
        phi[i, j, k] = float((d2 < R * R).mean())
    return phi


def test_hashin_demo_phase_fractions_and_mean_stress():
    """demo/elasticity/hashin/project.xml as written: place_fiber R=0.2 / R=0.4 at n = 64 with the reference's defaults
    smooth_levels = -1, smooth_tol = 1e-3 (F:14842-14843, doc/fileformat.xml:139-140).

    What the reference's current code computes there (F:17489-17581, F:16622-16752): every voxel within half a voxel
    diagonal of a sphere gets a volume FRACTION -- error estimate (r K)^2 = 4.6e-3 (R = 0.2) / 1.1e-3 (R = 0.4) >=
    smooth_tol at voxel level, 7.2e-4 / 1.8e-4 < smooth_tol one level down, so: eight sub-voxels, tangent-plane cuts.
    Binary (voxel-centre) fractions cannot come out of that code for any smooth_levels.

    Evidence that the smoothed fractions -- not a voxeliser error -- are what moves <sigma> from the demo's recorded
    12.9152 to 12.9203: the checker's fractions agree with a brute-force 32^3 super-sampling of the two spheres
    (an algorithm that shares nothing with the plane cuts) to 5e-3 per voxel and 3e-5 in the phase volumes (tangent
    planes lie outside a convex body: +1e-4 relative), and the oracle gives the same <sigma> = 12.920 for both, while binary fractions give the recorded 12.9152
    (test_oracle_pins.py::test_hashin_coated_sphere_demo_value).  The neutral-inclusion theory value is 12.91603:
    Voigt mixing in the interface voxels is an upper bound, hence the over-estimate with fractions.  The demo comment
    therefore records a binary voxelisation (older code); the XML as written yields 12.9203 in the current reference."""
    n = 64
    fibers = [Fiber("capsule", [.5, .5, .5], [1, 0, 0], 0.0, 0.2, 2), Fiber("capsule", [.5, .5, .5], [1, 0, 0], 0.0, 0.4, 1)]
    phi, _, _ = voxel_oracle.voxelize(fibers, (n, n, n), (1, 1, 1), (0, 0, 0), 3, 0)
    bf = np.stack([np.ones((n, n, n)), _supersampled_sphere(n, 0.4, 32), _supersampled_sphere(n, 0.2, 32)])
    for m, R in ((1, 0.4), (2, 0.2)):
        assert np.abs(phi[m] - bf[m]).max() < 6e-3                       # per voxel: the smoothing tolerance
        assert 0 <= phi[m].mean() - bf[m].mean() < 5e-5                  # phase volume: tangent planes over-estimate
        assert abs(bf[m].mean() - 4 / 3 * math.pi * R ** 3) < 2e-6      # the brute force is the sphere's volume
        frac = ((phi[m] > 0) & (phi[m] < 1)).mean()
        assert frac > 0.5 * 4 * math.pi * R ** 2 / n                     # a full shell of fractional voxels, not binary
    mats = [(1.0, 3.63867684478), (3.0, 2.0), (5.0, 4.0)]
    res = {}
    for name, f in (("checker", phi), ("brute force", bf)):
        o = LSOracle(n, n, n, mats=mats, phis=list(voxel_oracle.normalize_phi(f)), tol=1e-8)
        assert o.run([1, 1, 1, 0, 0, 0]) is False
        res[name] = o.mean_stress()[:3]
    assert np.abs(res["checker"] - 12.9203).max() < 2e-4
    assert np.abs(res["brute force"] - res["checker"]).max() < 2e-4
    assert np.abs(res["checker"] - 12.9152).min() > 4e-3                 # not the demo comment's (binary) value


def test_reference_self_tests_of_the_plane_box_cut(tmp_path):
    """The reference's own known answers for halfspace_box_cut_volume (F:1385-1577), 'halfspace cutting II' and 'III' of
    run_tests (F:23829-23862), on the checker's restatement AND on the product's closed form (fg_plane_cut.h compiled for the
    host): II -- axis-aligned planes through a 1 x 2 x 3 box cut off min(max(0, t), dim_j) * area; III -- mirroring the box,
    the plane and the normal in one axis leaves the cut volume unchanged."""
    import ctypes
    import os
    import subprocess
    from oracle import voxel_oracle
    lib = voxel_oracle.load()
    lib.ref_box_cut_volume.restype = ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    lib.ref_box_cut_volume.argtypes = [dp, dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_double]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "emu_cut.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DFG_HOST_EMULATION", "-ffp-contract=off", "-shared", "-fPIC", "-o", so,
                           os.path.join(root, "tests", "emulate", "emu_cut.cpp")])
    emu = ctypes.CDLL(so)
    emu.emu_box_fraction.restype = ctypes.c_double
    emu.emu_box_fraction.argtypes = [dp, dp, dp]
    P = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(dp)
    dim = np.array([1.0, 2.0, 3.0])
    vol = dim.prod()

    def both(x, n, x0):
        v_ref = lib.ref_box_cut_volume(P(x), P(n), P(x0), *dim)
        v_new = emu.emu_box_fraction(P(np.asarray(x) - np.asarray(x0)), P(n), P(dim)) * vol   # plane point relative to the box vertex
        return v_ref, v_new
    tol = 10 * np.finfo(float).eps * vol     # check_tol: sqrt(eps) in the reference; both forms are far better
    for k in range(-10, 30):                  # halfspace cutting II
        for j in range(3):
            t = dim[j] * k / 30.0
            n, x0 = np.zeros(3), np.zeros(3)
            n[j] = 1.0
            x0[j] = -t
            want = min(max(0.0, t), dim[j]) * dim[(j + 1) % 3] * dim[(j + 2) % 3]
            for v in both(np.zeros(3), n, x0):
                assert abs(v - want) < tol
    for k in range(-10, 30):                  # halfspace cutting III
        for j in range(3):
            n = np.full(3, 1 / np.sqrt(3))
            x0 = np.full(3, -2.0 * k / 30.0)
            a = both(np.zeros(3), n, x0)
            x0[j] *= -1
            n[j] *= -1
            x = np.zeros(3)
            x[j] += dim[j]
            b = both(x, n, x0)
            assert abs(a[0] - b[0]) < tol and abs(a[1] - b[1]) < tol and abs(a[0] - a[1]) < 1e-13 * vol
