"""Geometry pre-processing of the project layer: shapes placed with <place_fiber> ->
phase volume fractions + interface normals, through the GPU voxeliser fg_voxelize
(fibergen_amd/csrc/fg_voxelize.hip).  No CPU fallback: without a GPU it fails loudly."""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

KINDS = {"capsule": 0, "halfspace": 1}


def voxelize(fibers, shape, dims, x0, nphases, matrix_mat, want_normals=False, smooth_levels=-1, smooth_tol=1e-3,
             device=0):
    """Returns (phi[nphases,nx,ny,nz] before normalisation, normals[3,...] or None, {material: real volume fraction})."""
    lib = _lib.load()
    nx, ny, nz = shape
    arr = (_lib.FgFiber * max(len(fibers), 1))()
    for i, f in enumerate(fibers):
        arr[i].kind = KINDS[f.kind]
        arr[i].material = int(f.material)
        for k in range(3):
            arr[i].c[k] = float(f.c[k])
            arr[i].a[k] = float(f.a[k])
        arr[i].L = float(f.L)
        arr[i].R = float(f.R)
    phi = np.zeros((nphases, nx, ny, nz))
    normals = np.zeros((3, nx, ny, nz)) if want_normals else None
    real = np.zeros(nphases)
    x0a = np.asarray(x0, dtype=np.float64)
    err = ctypes.create_string_buffer(512)
    rc = lib.fg_voxelize(arr, len(fibers), nx, ny, nz, float(dims[0]), float(dims[1]), float(dims[2]),
                         x0a.ctypes.data_as(_lib.c_double_p), nphases, int(matrix_mat), int(smooth_levels),
                         float(smooth_tol), phi.ctypes.data_as(_lib.c_double_p),
                         normals.ctypes.data_as(_lib.c_double_p) if normals is not None else None,
                         real.ctypes.data_as(_lib.c_double_p), int(device), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    vol = float(dims[0]) * float(dims[1]) * float(dims[2])
    return phi, normals, {m: real[m] / vol for m in range(nphases)}
