"""The product voxeliser (GPU: fibergen_amd/csrc/fg_voxelize.hip -- stack-walked octree, closed-form plane / box
volumes in double-double) against its checker (oracle/c/fg_voxel_ref.cpp: the reference's recursive algorithm with the
polyhedron cut, F:16622-16752, F:1385-1577) and against known answers."""
import math

import numpy as np
import pytest

from oracle import voxel_oracle

pytestmark = pytest.mark.gpu

TOL = 4e-14   # volume fractions: two different algorithms for the same definition (the checker sums the polyhedron's
              # faces in plain double, the product's polynomial numerator is double-double); observed <= 1.3e-14


class Fiber:
    def __init__(self, kind, c, a, L, R, material):
        self.kind, self.c, self.a, self.L, self.R, self.material = kind, c, a, L, R, material


def both(fibers, shape, dims=(1, 1, 1), x0=(0, 0, 0), nph=2, matrix=0, **kw):
    from fibergen_amd import geometry
    got = geometry.voxelize(fibers, shape, dims, x0, nph, matrix, want_normals=True, **kw)
    want = voxel_oracle.voxelize(fibers, shape, dims, x0, nph, matrix, want_normals=True, **kw)
    return got, want


def random_capsules(K, seed, materials=(1,)):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(K):
        a = rng.standard_normal(3)
        out.append(Fiber("capsule", rng.random(3).tolist(), a.tolist(), float(rng.uniform(0.0, 0.5)), float(rng.uniform(0.03, 0.12)),
                         int(materials[i % len(materials)])))
    return out


CASES = {
    "sphere": dict(fibers=[Fiber("capsule", [.5, .5, .5], [1, 0, 0], 0.0, 0.3, 1)], shape=(32, 32, 32)),
    "capsule z": dict(fibers=[Fiber("capsule", [.5, .5, .5], [0, 0, 1], 0.6, 0.2, 1)], shape=(32, 32, 32)),
    "oblique capsule, anisotropic cell, offset origin":
        dict(fibers=[Fiber("capsule", [.1, .4, -.2], [1, 2, -1], 0.7, 0.11, 1)], shape=(24, 20, 36), dims=(1.0, 2.0, 1.5),
             x0=(-0.5, -0.6, -1.0)),
    "overlapping capsules, two materials": dict(fibers=random_capsules(12, 1, (1, 2)), shape=(40, 40, 40), nph=3),
    "many capsules": dict(fibers=random_capsules(60, 2), shape=(48, 48, 48)),
    "half spaces (laminate demo)":
        dict(fibers=[Fiber("halfspace", [0.0, .5, .5], [1, 0, 0], 0, 0.25, 0), Fiber("halfspace", [0.2, .5, .5], [-1, 0, 0], 0, 0.25, 1),
                     Fiber("halfspace", [0.5, .5, .5], [-1, 0, 0], 0, 0.25, 2)], shape=(10, 1, 1), nph=3),
    "oblique half space": dict(fibers=[Fiber("halfspace", [.5, .5, .5], [1, 1, 0], 0, 0.25, 1)], shape=(8, 8, 2)),
    "general half space + sphere": dict(fibers=[Fiber("halfspace", [.3, .4, .5], [0.3, -1.0, 0.45], 0, 0.25, 1),
                                                Fiber("capsule", [.6, .5, .4], [0, 0, 1], 0.0, 0.2, 1)], shape=(16, 12, 20)),
    "tiny sphere (deep refinement)": dict(fibers=[Fiber("capsule", [.52, .47, .5], [0, 0, 1], 0.0, 0.02, 1)], shape=(16, 16, 16)),
    "hashin coated sphere": dict(fibers=[Fiber("capsule", [.5, .5, .5], [1, 0, 0], 0.0, 0.2, 2),
                                         Fiber("capsule", [.5, .5, .5], [1, 0, 0], 0.0, 0.4, 1)], shape=(64, 64, 64), nph=3),
}


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("levels", [-1, 0, 2])
def test_product_equals_checker(name, levels):
    case = dict(CASES[name])
    fibers, shape = case.pop("fibers"), case.pop("shape")
    (phi, nrm, real), (phi_c, nrm_c, real_c) = both(fibers, shape, smooth_levels=levels, **case)
    assert np.abs(phi - phi_c).max() <= TOL
    assert np.abs(voxel_oracle.normalize_phi(phi) - voxel_oracle.normalize_phi(phi_c)).max() <= TOL   # F:17613-17626 order
    assert np.abs(nrm - nrm_c).max() <= 1e-13
    assert real == real_c
    assert ((phi >= 0) & (phi <= 1)).all()


def test_smooth_tol_and_known_answers():
    sph = [Fiber("capsule", [.5, .5, .5], [1, 0, 0], 0.0, 0.3, 1)]
    exact = 4 / 3 * math.pi * 0.3 ** 3
    errs = []
    for tol in (1e-1, 1e-3, 1e-5):
        (phi, nrm, real), (phi_c, _, _) = both(sph, (32, 32, 32), smooth_tol=tol)
        assert np.abs(phi - phi_c).max() <= TOL
        errs.append(abs(phi[1].mean() - exact) / exact)
    assert errs[2] < errs[1] < errs[0] and errs[2] < 2e-5      # the tolerance drives the refinement
    assert real[1] == pytest.approx(exact, rel=1e-14)
    assert np.allclose((nrm * nrm).sum(axis=0), 1.0) and nrm[0, 31, 16, 16] > 0.99 and nrm[0, 0, 16, 16] < -0.99
    # a plane cutting voxels obliquely: exact volume fraction of the half space x + y < 1 is 1/2
    from fibergen_amd import geometry
    phi, _, _ = geometry.voxelize([Fiber("halfspace", [.5, .5, .5], [1, 1, 0], 0, 0.25, 1)], (8, 8, 2), (1, 1, 1), (0, 0, 0), 2, 0)
    assert phi[1].mean() == pytest.approx(0.5, abs=1e-15)
    # nearly axis-aligned planes: the closed form stays exact where a normal component (almost) vanishes
    for tilt in (0.0, 1e-17, 1e-12, 1e-9, 1e-6, 1e-3):
        f = [Fiber("halfspace", [.437, .5, .5], [1.0, tilt, -2 * tilt], 0, 0.25, 1)]
        (phi, _, _), (phi_c, _, _) = both(f, (10, 6, 4))
        assert np.abs(phi - phi_c).max() <= TOL, tilt
    with pytest.raises(RuntimeError, match="zero normal"):
        geometry.voxelize([Fiber("halfspace", [0, 0, 0], [0, 0, 0], 0, 0.1, 1)], (4, 4, 4), (1, 1, 1), (0, 0, 0), 2, 0)
    with pytest.raises(RuntimeError, match="orientation"):
        geometry.voxelize([Fiber("capsule", [0, 0, 0], [0, 0, 0], 0.5, 0.1, 1)], (4, 4, 4), (1, 1, 1), (0, 0, 0), 2, 0)


def test_bench_scale_geometry():
    """256^3 with the 126 capsules of the benchmark family through <place_fiber>'s voxeliser: seconds on the GPU"""
    import time
    from fibergen_amd import geometry
    from fibergen_amd.rve import bench_rve_parameters, place_capsules
    par = bench_rve_parameters(256)
    centres, axes = place_capsules(par["K"], par["R"], par["L"])
    # <place_fiber L=...> is the equal-volume length: cylinder part L_total - 2R = L - 4/3 R
    fibers = [Fiber("capsule", c.tolist(), a.tolist(), float(par["L"] - 2 * par["R"] + 4 / 3 * par["R"]), float(par["R"]), 1)
              for c, a in zip(centres, axes)]
    t0 = time.perf_counter()
    phi, nrm, real = geometry.voxelize(fibers, (256, 256, 256), (1, 1, 1), (0, 0, 0), 2, 0, want_normals=True)
    dt = time.perf_counter() - t0
    inside = [f for f in fibers if all(0.2 < x < 0.8 for x in f.c)]     # (shapes are not periodic in <place_fiber>)
    assert dt < 60
    assert phi[1].max() == 1.0 and 0.02 < ((phi[1] > 0) & (phi[1] < 1)).mean() < 0.05
    assert len(inside) > 5 and phi[1].mean() < real[1] + 1e-3


@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_equal_checker(seed):
    """Seeded random scenes: grid and cell shape, origin, a mix of spheres, capsules and half spaces of up to three
    materials, overlapping, any smoothing depth / tolerance."""
    rng = np.random.default_rng(300 + seed)
    shape = tuple(int(v) for v in rng.choice([1, 2, 5, 8, 12, 16, 24, 33], size=3))
    dims = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3))
    x0 = tuple(float(v) for v in rng.uniform(-1.0, 1.0, size=3))
    nmat = int(rng.integers(1, 4))
    fibers = []
    for _ in range(int(rng.integers(1, 16))):
        c = [x0[k] + rng.random() * dims[k] for k in range(3)]
        a = rng.standard_normal(3).tolist()
        m = int(rng.integers(1, nmat + 1))
        kind = rng.random()
        if kind < 0.15:
            fibers.append(Fiber("halfspace", c, a, 0, 0.25, m))
        elif kind < 0.4:
            fibers.append(Fiber("capsule", c, a, 0.0, float(rng.uniform(0.02, 0.4)), m))
        else:
            fibers.append(Fiber("capsule", c, a, float(rng.uniform(0.0, 0.8)), float(rng.uniform(0.02, 0.2)), m))
    kw = dict(smooth_levels=int(rng.choice([-1, 0, 1, 2])), smooth_tol=float(10.0 ** rng.uniform(-4, -1)))   # (24 seeds with levels 3 / tol 1e-5 passed too: 2 min of checker time)
    (phi, nrm, real), (phi_c, nrm_c, real_c) = both(fibers, shape, dims, x0, nph=nmat + 1, **kw)
    assert np.abs(phi - phi_c).max() <= TOL, (seed, shape, kw)
    assert np.abs(nrm - nrm_c).max() <= 1e-13, (seed, shape, kw)
    assert real == real_c


@pytest.mark.parametrize("depth", [1, 2, 3])
@pytest.mark.parametrize("scene", ["fine", "coarse"])
def test_team_refinement_equals_one_thread_walk(scene, depth, monkeypatch):
    """Grids with few interface voxels give each of them 8^depth threads (k_vox_refine_team): the children's volumes are
    added in child order and clipped per node as in the one-thread walk, so the fractions are bit-identical -- a voxel's
    value does not depend on how many interface voxels its grid has."""
    from fibergen_amd import geometry
    if scene == "fine":
        args = (random_capsules(12, 1, (1, 2)), (24, 20, 28), (1.0, 1.3, 0.9), (0.1, -0.2, 0.0), 3, 0)
        kws = [dict(smooth_levels=-1, smooth_tol=1e-3), dict(smooth_levels=2, smooth_tol=0.0), dict(smooth_levels=4, smooth_tol=0.0)]
    else:   # voxels larger than the shapes: deep trees, closed nodes right below the root
        args = (random_capsules(15, 7), (6, 1, 2), (0.9, 1.7, 1.2), (0.0, 0.0, 0.0), 2, 0)
        kws = [dict(smooth_levels=-1, smooth_tol=2e-3), dict(smooth_levels=1, smooth_tol=0.0)]
    for kw in kws:
        from fibergen_amd import _lib
        lib = _lib.load()
        try:
            lib.fg_voxelize_team_depth(0)
            want = geometry.voxelize(*args, want_normals=False, **kw)[0]
            lib.fg_voxelize_team_depth(depth)
            got = geometry.voxelize(*args, want_normals=False, **kw)[0]
        finally:
            lib.fg_voxelize_team_depth(-1)
        assert ((want > 0) & (want < 1)).any()
        assert np.array_equal(got, want), (scene, depth, kw)
