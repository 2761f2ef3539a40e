"""Randomised combinations through the C ABI against the NumPy oracle: every seed draws a grid (powers of two, odd,
prime, p * 2^k lengths, tile-sized rows), a cell, one to three phases with smooth random fraction fields and random
isotropic moduli, the mixing rule, the Green operator, the solution method, the loop variant (u_loop / u_tile / fuse_x),
optionally mixed boundary conditions and load steps -- the combinations the hand-written cases of the other files do not
enumerate.  The draw is deterministic per seed; a failing seed is a bug report.

Bars: iteration counts equal, residual histories 1e-9, strain fields 1e-8, mean stress 1e-9 (runs to tol 1e-7)."""
import os

import numpy as np
import pytest

from helpers import lame, rel_err

# FG_FUZZ_SEEDS=n widens both sweeps to n seeds (a soak run; the suite keeps the first 150 / 60)
N_SINGLE = int(os.environ.get("FG_FUZZ_SEEDS", "150"))
N_SLAB = int(os.environ.get("FG_FUZZ_SEEDS", "60"))

pytestmark = pytest.mark.gpu

LENGTHS_XY = [1, 2, 4, 5, 6, 7, 8, 10, 12, 16, 20, 24]
LENGTHS_Z = [1, 4, 5, 6, 8, 9, 12, 16, 24, 124, 128]
PROJECTORS = {   # diagonal of the Voigt projector onto the prescribed-strain components (shear entries carry 1/2); the rest: stress
    "uniaxial_stress_x": [1, 0, 0, 0, 0, 0],
    "plane_strain_free_z": [1, 1, 0, 0.5, 0.5, 0.5],
    "shear_only": [0, 0, 0, 0, 0, 0.5],
}


def smooth_field(rng, shape):
    """a smooth periodic field in [0, 1] with flat parts at both ends (pure voxels and mixtures)"""
    x = [np.arange(m) / m for m in shape]
    f = np.zeros(shape)
    for _ in range(3):
        k = rng.integers(0, 3, size=3)
        ph = rng.uniform(0, 2 * np.pi, size=3)
        f += rng.uniform(0.5, 1.0) * (np.cos(2 * np.pi * k[0] * x[0] + ph[0])[:, None, None] *
                                      np.cos(2 * np.pi * k[1] * x[1] + ph[1])[None, :, None] *
                                      np.cos(2 * np.pi * k[2] * x[2] + ph[2])[None, None, :])
    f = (f - f.min()) / max(f.max() - f.min(), 1e-300)
    return np.clip(1.6 * f - 0.3, 0.0, 1.0)


# lengths the Stockham tile kernels take (fg_fft_smooth.h; most of them in the per-plan tables of fg_fft_smooth_plans.h, 108 / 130 /
# 162 / 250 with the class kernels or odd rows), drawn as ONE long axis of a thin grid
TILE_LENGTHS = [100, 108, 112, 120, 144, 150, 160, 192, 200, 224, 240, 250, 300, 320, 384, 400]
TILE_LENGTHS_Z = [100, 130, 144, 162, 200, 225, 250, 288, 320, 400]
N_TILE = int(os.environ.get("FG_FUZZ_SEEDS", "36"))


def draw(seed, tile_lengths=False):
    rng = np.random.default_rng((9000 if tile_lengths else 1000) + seed)
    while True:
        if tile_lengths:
            axis = int(rng.integers(0, 3))
            shape = [int(rng.choice([2, 4, 5, 6, 8])), int(rng.choice([2, 4, 5, 6, 8])), int(rng.choice([4, 6, 8, 10]))]
            shape[axis] = int(rng.choice(TILE_LENGTHS_Z if axis == 2 else TILE_LENGTHS))
            shape = tuple(shape)
            break
        shape = (int(rng.choice(LENGTHS_XY)), int(rng.choice(LENGTHS_XY)), int(rng.choice(LENGTHS_Z)))
        if 8 <= shape[0] * shape[1] * shape[2] <= 40000:
            break
    dims = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3))
    nph = int(rng.integers(1, 4))
    mixing = "laminate" if (nph == 2 and rng.random() < 0.5) else "voigt"
    mats = [lame(E=float(rng.uniform(0.5, 20.0)), nu=float(rng.uniform(0.05, 0.4))) for _ in range(nph)]
    if nph == 1:
        phis = [np.ones(shape)]
    elif nph == 2:
        p1 = smooth_field(rng, shape)
        phis = [1.0 - p1, p1]
    else:
        p1, p2 = smooth_field(rng, shape), smooth_field(rng, shape)
        p2 = np.minimum(p2, 1.0 - p1)
        phis = [1.0 - p1 - p2, p1, p2]
    normals = None
    if mixing == "laminate":
        v = rng.standard_normal((3,) + shape)
        normals = v / np.sqrt((v * v).sum(axis=0))
    scheme = "collocated" if rng.random() < 0.2 else "staggered"
    method = "cg" if rng.random() < 0.35 else "basic"
    opts = {}
    if rng.random() < 0.5:
        opts["u_loop"] = int(rng.integers(0, 3))
    if rng.random() < 0.3:
        opts["fuse_x"] = int(rng.integers(0, 2))
    if rng.random() < 0.3:
        opts["u_tile"] = int(rng.choice([0, 1]))
    if rng.random() < 0.3:
        opts["fuse_stress_div"] = int(rng.integers(0, 2))
    bc = None
    if rng.random() < 0.3:
        bc = str(rng.choice(list(PROJECTORS)))
    steps = None
    if rng.random() < 0.25:
        steps = [0.0, 0.4, 1.0]
    E = rng.uniform(-1.0, 1.0, size=6)
    return dict(shape=shape, dims=dims, mats=mats, phis=phis, normals=normals, mixing=mixing, scheme=scheme, method=method,
                opts=opts, bc=bc, steps=steps, E=E)


@pytest.mark.parametrize("seed", range(N_TILE))
def test_random_combination_on_tile_kernel_lengths_matches_oracle(seed):
    """the same draw with one axis of a length the Stockham tile kernels transform (decimal and p * 2^k lengths from 100 points
    on, per-plan and class kernels, fused x pass, odd rows)"""
    _run_combination(seed, True)


@pytest.mark.parametrize("seed", range(N_SINGLE))
def test_random_combination_matches_oracle(seed):
    _run_combination(seed, False)


def _run_combination(seed, tile_lengths):
    from fibergen_amd import LSSolver
    from oracle.ls_oracle import LSOracle
    c = draw(seed, tile_lengths)
    shape, dims = c["shape"], c["dims"]
    common = dict(tol=1e-7, maxiter=400)
    o = LSOracle(*shape, *dims, mats=c["mats"], phis=c["phis"], normals=c["normals"], mixing_rule=c["mixing"],
                 gamma_scheme=c["scheme"], **common)
    s = LSSolver(*shape, *dims)
    s.set_num_phases(len(c["mats"]))
    for p, (m, phi) in enumerate(zip(c["mats"], c["phis"])):
        s.set_phase(p, m[0], m[1], phi)
    if c["normals"] is not None:
        s.set_normals(c["normals"])
    s.set_options(mixing_rule=c["mixing"], gamma_scheme=c["scheme"], method=c["method"], **common, **c["opts"])
    E, S0, P = c["E"].copy(), np.zeros(6), None
    if c["bc"] is not None:
        keep = np.array(PROJECTORS[c["bc"]], dtype=float)
        P = np.diag(keep)
        E = E * (keep > 0)                 # prescribed strain lives in the range of P, the stress (zero) in its complement
        s.set_bc_projector(P)
    params = c["steps"] or [0.0, 1.0]
    tag = "seed %d: %s" % (seed, {k: c[k] for k in ("shape", "mixing", "scheme", "method", "opts", "bc", "steps")})
    try:
        ref_failed = o.run_load_steps(E, S0, P, params=params, method=c["method"])
    except RuntimeError as e:
        # a combination the reference rejects must be rejected by the product with the same message
        with pytest.raises(RuntimeError) as got:
            s.run_load_steps(E, S0, params=params)
        assert str(e).split(":")[0][:24] in str(got.value), tag
        s.close()
        return
    failed = s.run_load_steps(E, S0, params=params)
    assert failed == ref_failed, tag
    assert s.iterations == o.iterations, tag
    r, rr = np.array(s.residuals), np.array(o.residuals)
    assert r.shape == rr.shape and np.abs(r - rr).max() < 1e-9, tag
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8, tag
    assert np.abs(s.mean_stress() - o.mean_stress()).max() < 1e-9 * max(1.0, np.abs(o.mean_stress()).max()), tag
    s.close()


def draw_slab(seed):
    rng = np.random.default_rng(5000 + seed)
    P = int(rng.choice([1, 2, 4]))
    fast = rng.random() < 0.5               # grids the tiled sweep fits (displacement loop per slab) or small / odd ones
    if fast:
        shape = (P * int(rng.choice([4, 8])), int(rng.choice([16, 32])) if P < 4 else 16, int(rng.choice([124, 128])))
    else:
        while True:
            shape = (P * int(rng.choice([1, 2, 3, 4, 6])), P * int(rng.choice([1, 2, 3, 4, 5])), int(rng.choice(LENGTHS_Z[:9])))
            if 8 <= shape[0] * shape[1] * shape[2] <= 20000:
                break
    dims = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3))
    nph = int(rng.integers(1, 4))
    mixing = "laminate" if (nph == 2 and rng.random() < 0.6) else "voigt"
    mats = [lame(E=float(rng.uniform(0.5, 20.0)), nu=float(rng.uniform(0.05, 0.4))) for _ in range(nph)]
    if nph == 1:
        phis = [np.ones(shape)]
    elif nph == 2:
        p1 = smooth_field(rng, shape)
        phis = [1.0 - p1, p1]
    else:
        p1, p2 = smooth_field(rng, shape), smooth_field(rng, shape)
        p2 = np.minimum(p2, 1.0 - p1)
        phis = [1.0 - p1 - p2, p1, p2]
    v = rng.standard_normal((3,) + shape)
    normals = v / np.sqrt((v * v).sum(axis=0))
    bc = str(rng.choice(list(PROJECTORS))) if rng.random() < 0.3 else None
    opts = {"slab_split": int(rng.integers(0, 2))}
    if rng.random() < 0.3:
        opts["fuse_x"] = int(rng.integers(0, 2))
    if rng.random() < 0.2:
        opts["phi_sweep"] = 0
    # round 3: CG and load steps on the slabs (drawn last, so the earlier draws of a seed stay what they were)
    method = "cg" if rng.random() < 0.35 else "basic"
    steps = [0.0, 0.4, 1.0] if rng.random() < 0.3 else None
    return dict(P=P, shape=shape, dims=dims, mats=mats, phis=phis, normals=normals, mixing=mixing, bc=bc, opts=opts,
                E=rng.uniform(-1.0, 1.0, size=6), method=method, steps=steps)


@pytest.mark.parametrize("seed", range(N_SLAB))
def test_random_slab_group_matches_oracle(seed):
    """The slab driver (all P slabs in this process, exchanges as device copies) on random problems."""
    from fibergen_amd.distributed import SlabGroup
    from oracle.ls_oracle import LSOracle
    c = draw_slab(seed)
    shape, dims = c["shape"], c["dims"]
    common = dict(tol=1e-7, maxiter=400)
    o = LSOracle(*shape, *dims, mats=c["mats"], phis=c["phis"], normals=c["normals"], mixing_rule=c["mixing"], **common)
    g = SlabGroup(*shape, *dims, nranks=c["P"])
    g.set_num_phases(len(c["mats"]))
    for p, (m, phi) in enumerate(zip(c["mats"], c["phis"])):
        g.set_phase(p, m[0], m[1], phi)
    g.set_normals(c["normals"])
    g.set_options(mixing_rule=c["mixing"], method=c["method"], **common, **c["opts"])
    E, S0, P = c["E"].copy(), np.zeros(6), None
    if c["bc"] is not None:
        keep = np.array(PROJECTORS[c["bc"]], dtype=float)
        P = np.diag(keep)
        E = E * (keep > 0)
        g.set_bc_projector(P)
    tag = "seed %d: %s" % (seed, {k: c[k] for k in ("P", "shape", "mixing", "opts", "bc", "method", "steps")})
    if c["steps"] is not None:
        assert o.run_load_steps(E, S0, P, params=c["steps"], method=c["method"]) is False, tag
        assert g.run_load_steps(E, S0, params=c["steps"]) is False, tag
    elif c["method"] == "cg":
        assert o.run_cg(E, S0, P) is False and g.run(E, S0) is False, tag
    else:
        assert o.run(E, S0, P) is False and g.run(E, S0) is False, tag
    assert g.iterations == o.iterations, tag
    assert np.abs(np.array(g.residuals) - np.array(o.residuals)).max() < 1e-9, tag
    assert rel_err(g.get_field("epsilon"), o.eps) < 1e-8, tag
    # (CG stops on a norm difference: two runs that agree to 1e-8 in the field agree to that, not better, in its means)
    mtol = 1e-8 if c["method"] == "cg" else 1e-9
    assert np.abs(g.mean_stress() - o.mean_stress()).max() < mtol * max(1.0, np.abs(o.mean_stress()).max()), tag
    g.close()


def draw_scalar(seed):
    rng = np.random.default_rng(9000 + seed)
    while True:
        shape = (int(rng.choice(LENGTHS_XY)), int(rng.choice(LENGTHS_XY)), int(rng.choice(LENGTHS_Z)))
        if 8 <= shape[0] * shape[1] * shape[2] <= 40000:
            break
    dims = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3))
    nph = int(rng.integers(1, 4))
    mus = [float(rng.uniform(0.05, 20.0)) for _ in range(nph)]
    if nph == 1:
        phis = [np.ones(shape)]
    elif nph == 2:
        p1 = smooth_field(rng, shape)
        phis = [1.0 - p1, p1]
    else:
        p1, p2 = smooth_field(rng, shape), smooth_field(rng, shape)
        p2 = np.minimum(p2, 1.0 - p1)
        phis = [1.0 - p1 - p2, p1, p2]
    return dict(shape=shape, dims=dims, mus=mus, phis=phis, mode=str(rng.choice(["heat", "porous", "viscosity"])),
                method="cg" if rng.random() < 0.35 else "basic", u_loop=int(rng.choice([1, 2])),
                bc=rng.random() < 0.25, E=rng.uniform(-1.0, 1.0, size=6))


@pytest.mark.parametrize("seed", range(N_SLAB))
def test_random_scalar_and_viscosity_problems_match_their_oracles(seed):
    """mode = heat / porous (potential-based loop, ScalarOracle) and viscosity (dual Stokes scheme, ViscosityOracle)."""
    from fibergen_amd import LSSolver
    c = draw_scalar(seed)
    shape, dims, mode = c["shape"], c["dims"], c["mode"]
    common = dict(tol=1e-7, maxiter=300)
    s = LSSolver(*shape, *dims)
    s.set_options(mode=mode)
    s.set_num_phases(len(c["mus"]))
    for p, (mu, phi) in enumerate(zip(c["mus"], c["phis"])):
        s.set_phase(p, mu, 0.0, phi)
    tag = "seed %d: %s" % (seed, {k: c[k] for k in ("shape", "mode", "method", "u_loop", "bc")})
    if mode == "viscosity":
        from oracle.viscosity_oracle import ViscosityOracle
        o = ViscosityOracle(*shape, *dims, mats=[(m, 0.0) for m in c["mus"]], phis=c["phis"], **common)
        E = c["E"].copy()
        E[:3] -= E[:3].mean()               # the prescribed fluid stress is traceless
        s.set_options(method=c["method"], **common)
        if c["bc"]:                          # round 3: shear stresses prescribed, normal shear rates zero
            P6 = np.diag([0.0, 0.0, 0.0, 0.5, 0.5, 0.5])
            E = E * np.array([0, 0, 0, 1.0, 1.0, 1.0])
            s.set_bc_projector(P6)
            s.set_options(bc_tol=1e-8)
            o.bc_tol = 1e-8
            ref_failed = o.run_cg(E, np.zeros(6), P6) if c["method"] == "cg" else o.run(E, np.zeros(6), P6)
            failed = s.run(E, np.zeros(6))
        else:
            ref_failed = o.run_cg(E) if c["method"] == "cg" else o.run(E)
            failed = s.run(E)
    else:
        from oracle.scalar_oracle import ScalarOracle
        o = ScalarOracle(*shape, mus=c["mus"], phis=c["phis"], dx=dims[0], dy=dims[1], dz=dims[2], **common)
        E = c["E"][:3].copy()
        s.set_options(method=c["method"], u_loop=c["u_loop"], **common)
        if c["bc"] and c["method"] == "basic":
            P3 = np.diag([1.0, 0.0, 1.0])   # gradient prescribed along x and z, zero mean flux along y
            P6 = np.zeros((6, 6))
            P6[:3, :3] = P3
            E = E * np.array([1.0, 0.0, 1.0])
            s.set_bc_projector(P6)
            s.set_options(bc_tol=1e-8)
            o.bc_tol = 1e-8
            ref_failed = o.run(E, np.zeros(3), P3)
            failed = s.run(E, np.zeros(6))
        else:
            ref_failed = o.run_cg(E) if c["method"] == "cg" else o.run(E)
            failed = s.run(E)
    assert failed == ref_failed, tag
    assert s.iterations == o.iterations, tag
    assert np.abs(np.array(s.residuals) - np.array(o.residuals)).max() < 1e-9, tag
    assert rel_err(s.get_field("epsilon"), o.eps) < 1e-8, tag
    ms = np.asarray(s.mean_stress())[:len(np.atleast_1d(o.mean_stress()))]
    assert np.abs(ms - o.mean_stress()).max() < 1e-9 * max(1.0, np.abs(o.mean_stress()).max()), tag
    s.close()


def draw_scalar_slab(seed):
    rng = np.random.default_rng(13000 + seed)
    P = int(rng.choice([1, 2, 3, 4]))
    if rng.random() < 0.3:                  # grids the tiled sweep fits
        shape = (P * int(rng.choice([4, 8])), P * int(rng.choice([4, 6])) if P > 1 else 16, int(rng.choice([124, 128])))
    else:
        while True:
            shape = (P * int(rng.choice([1, 2, 3, 4, 6])), P * int(rng.choice([1, 2, 3, 4, 5])), int(rng.choice(LENGTHS_Z[:9])))
            if 8 <= shape[0] * shape[1] * shape[2] <= 20000:
                break
    dims = tuple(float(v) for v in rng.uniform(0.5, 2.0, size=3))
    nph = int(rng.integers(1, 4))
    mus = [float(rng.uniform(0.05, 20.0)) for _ in range(nph)]
    if nph == 1:
        phis = [np.ones(shape)]
    elif nph == 2:
        p1 = smooth_field(rng, shape)
        phis = [1.0 - p1, p1]
    else:
        p1, p2 = smooth_field(rng, shape), smooth_field(rng, shape)
        p2 = np.minimum(p2, 1.0 - p1)
        phis = [1.0 - p1 - p2, p1, p2]
    opts = {}
    if rng.random() < 0.3:
        opts["fuse_x"] = int(rng.integers(0, 2))
    return dict(P=P, shape=shape, dims=dims, mus=mus, phis=phis, mode=str(rng.choice(["heat", "porous"])),
                method="cg" if rng.random() < 0.35 else "basic", bc=rng.random() < 0.3, E=rng.uniform(-1.0, 1.0, size=3), opts=opts)


@pytest.mark.parametrize("seed", range(N_SLAB))
def test_random_scalar_slab_group_matches_oracle(seed):
    """mode = heat / porous on the slab driver: any grid (tiled or untiled potential sweep with halo planes, fused or separate
    x pass on the y-slab), basic scheme (also with a prescribed mean flux) and CG."""
    from fibergen_amd.distributed import SlabGroup
    from oracle.scalar_oracle import ScalarOracle
    c = draw_scalar_slab(seed)
    shape, dims = c["shape"], c["dims"]
    common = dict(tol=1e-7, maxiter=300)
    g = SlabGroup(*shape, *dims, nranks=c["P"])
    g.set_options(mode=c["mode"])
    g.set_num_phases(len(c["mus"]))
    for p, (mu, phi) in enumerate(zip(c["mus"], c["phis"])):
        g.set_phase(p, mu, 0.0, phi)
    g.set_options(method=c["method"], **common, **c["opts"])
    o = ScalarOracle(*shape, mus=c["mus"], phis=c["phis"], dx=dims[0], dy=dims[1], dz=dims[2], **common)
    tag = "seed %d: %s" % (seed, {k: c[k] for k in ("P", "shape", "mode", "method", "bc", "opts")})
    E = c["E"].copy()
    if c["bc"] and c["method"] == "basic":
        P3 = np.diag([1.0, 0.0, 1.0])       # gradient prescribed along x and z, zero mean flux along y
        P6 = np.zeros((6, 6))
        P6[:3, :3] = P3
        E = E * np.array([1.0, 0.0, 1.0])
        g.set_bc_projector(P6)
        g.set_options(bc_tol=1e-8)
        o.bc_tol = 1e-8
        ref_failed = o.run(E, np.zeros(3), P3)
        failed = g.run(E, np.zeros(6))
    else:
        ref_failed = o.run_cg(E) if c["method"] == "cg" else o.run(E)
        failed = g.run(E)
    assert failed == ref_failed, tag
    assert g.iterations == o.iterations, tag
    assert np.abs(np.array(g.residuals) - np.array(o.residuals)).max() < 1e-9, tag
    assert rel_err(g.get_field("epsilon"), o.eps) < 1e-8, tag
    mtol = 1e-8 if c["method"] == "cg" else 1e-9
    assert np.abs(np.asarray(g.mean_stress())[:3] - o.mean_stress()).max() < mtol * max(1.0, np.abs(o.mean_stress()).max()), tag
    g.close()
