"""Deterministic synthetic fibre RVEs.

K capsules (sphero-cylinders) of radius R and cylinder length L, centres uniform in the
unit cell, axes uniform on the sphere, all from numpy.random.default_rng(seed); capsules
are periodic.  Volume fractions come from the signed distance d of the voxel centre to
the nearest capsule surface, phi = clamp(1/2 - d/h, 0, 1) with h the voxel size, and the
interface normal is the distance gradient of that capsule (pointing out of the inclusion,
as in the reference, F:5286-5294).  This is a data generator for benchmarks and tests, not
a restatement of the reference's adaptive voxeliser.
"""
from __future__ import annotations

import numpy as np


def _capsule_distance(px, py, pz, c, a, L, R):
    """Signed distance to a capsule with centre c, unit axis a; also the outward gradient."""
    dx, dy, dz = px - c[0], py - c[1], pz - c[2]
    t = dx * a[0] + dy * a[1] + dz * a[2]
    t = np.clip(t, -0.5 * L, 0.5 * L)
    qx, qy, qz = dx - t * a[0], dy - t * a[1], dz - t * a[2]
    r = np.sqrt(qx * qx + qy * qy + qz * qz)
    rs = np.where(r > 0, r, 1.0)
    return r - R, qx / rs, qy / rs, qz / rs


def synthetic_fiber_rve(n, K=40, R=0.05, L=0.4, seed=0, with_normals=True):
    """Returns (phi_inclusion [n,n,n], normals [3,n,n,n] or None) on the unit cube."""
    if isinstance(n, int):
        n = (n, n, n)
    nx, ny, nz = n
    rng = np.random.default_rng(seed)
    centres = rng.random((K, 3))
    v = rng.standard_normal((K, 3))
    axes = v / np.sqrt((v * v).sum(axis=1))[:, None]
    h = 1.0 / max(n)
    dist = np.full(n, np.inf)
    normals = np.zeros((3,) + n) if with_normals else None
    reach = 0.5 * L + R + 2 * h
    for c, a in zip(centres, axes):
        # periodic bounding box in index space
        lo = np.floor((c - reach) * np.array(n)).astype(int)
        hi = np.ceil((c + reach) * np.array(n)).astype(int)
        ix = np.arange(lo[0], hi[0] + 1)
        iy = np.arange(lo[1], hi[1] + 1)
        iz = np.arange(lo[2], hi[2] + 1)
        px = ((ix + 0.5) / nx)[:, None, None]
        py = ((iy + 0.5) / ny)[None, :, None]
        pz = ((iz + 0.5) / nz)[None, None, :]
        d, gx, gy, gz = _capsule_distance(px, py, pz, c, a, L, R)
        sel = np.ix_(ix % nx, iy % ny, iz % nz)
        cur = dist[sel]
        closer = d < cur
        dist[sel] = np.where(closer, d, cur)
        if with_normals:
            for comp, g in enumerate((gx, gy, gz)):
                cn = normals[comp][sel]
                normals[comp][sel] = np.where(closer, g, cn)
    phi = np.clip(0.5 - dist / h, 0.0, 1.0)
    return phi, normals
