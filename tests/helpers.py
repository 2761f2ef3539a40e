"""Shared test geometry / set-up helpers (CPU only, no product imports at module level)."""
import numpy as np


def sphere_phi(n, R=0.3, c=(0.5, 0.5, 0.5), sub=4):
    """Sub-sampled volume fraction of a sphere in the unit cell."""
    nx, ny, nz = n
    s = (np.arange(sub) + 0.5) / sub
    x = ((np.arange(nx)[:, None] + s[None, :]) / nx).reshape(-1)
    y = ((np.arange(ny)[:, None] + s[None, :]) / ny).reshape(-1)
    z = ((np.arange(nz)[:, None] + s[None, :]) / nz).reshape(-1)
    d2 = ((x[:, None, None] - c[0]) ** 2 + (y[None, :, None] - c[1]) ** 2 + (z[None, None, :] - c[2]) ** 2)
    inside = (d2 <= R * R).astype(np.float64)
    return inside.reshape(nx, sub, ny, sub, nz, sub).mean(axis=(1, 3, 5))


def sphere_normals(n, c=(0.5, 0.5, 0.5)):
    nx, ny, nz = n
    x = (np.arange(nx) + 0.5) / nx - c[0]
    y = (np.arange(ny) + 0.5) / ny - c[1]
    z = (np.arange(nz) + 0.5) / nz - c[2]
    v = np.stack(np.broadcast_arrays(x[:, None, None], y[None, :, None], z[None, None, :])).astype(np.float64)
    r = np.sqrt((v * v).sum(axis=0))
    r[r == 0] = 1.0
    return v / r


MATRIX = dict(E=1.0, nu=0.3)      # SURVEY 8d config 1 materials
INCLUSION = dict(E=10.0, nu=0.2)


def lame(E, nu):
    lam = (E * nu) / ((1 + nu) * (1 - 2 * nu))
    mu = E / (2 * (1 + nu))
    return mu, lam


def two_phase_setup(n, mixing="voigt", R=0.3):
    """(mats, phis, normals) for a centred sphere of the inclusion material."""
    phi = sphere_phi(n, R=R)
    mats = [lame(**MATRIX), lame(**INCLUSION)]
    return mats, [1.0 - phi, phi], sphere_normals(n)


def make_oracle(n, dims=(1.0, 1.0, 1.0), mixing="voigt", **kw):
    from oracle.ls_oracle import LSOracle
    mats, phis, normals = two_phase_setup(n, mixing)
    return LSOracle(*n, *dims, mats=mats, phis=phis, normals=normals, mixing_rule=mixing, **kw)


def make_gpu_solver(n, dims=(1.0, 1.0, 1.0), mixing="voigt", **kw):
    from fibergen_amd import LSSolver
    mats, phis, normals = two_phase_setup(n, mixing)
    s = LSSolver(*n, *dims)
    s.set_num_phases(2)
    for p in range(2):
        s.set_phase(p, mats[p][0], mats[p][1], phis[p])
    s.set_normals(normals)
    s.set_options(mixing_rule=mixing, **kw)
    return s


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    d = np.abs(a - b).max()
    s = np.abs(b).max()
    return d / s if s > 0 else d


def emulation_build_flags():
    """g++ flags of the host builds of the kernels' per-thread code (tests/emulate); FG_EMU_SANITIZE=1 adds
    AddressSanitizer + UBSan (the process must then run with libasan preloaded, see tests/test_emulation_sanitized.py)."""
    import os
    flags = ["-std=c++17", "-DFG_HOST_EMULATION", "-ffp-contract=off", "-shared", "-fPIC"]
    if os.environ.get("FG_EMU_SANITIZE") == "1":
        return ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] + flags
    return ["-O2"] + flags
