"""ctypes wrapper of oracle/c/libfg_voxel_ref.so: the CHECKER of the voxeliser -- the reference's recursive
integratePhiVoxel / halfspace_box_cut_volume (F:16622-16752, F:1385-1577) restated on the host.
TEST INFRASTRUCTURE ONLY (the product is fibergen_amd/csrc/fg_voxelize.hip on the GPU)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = ctypes.POINTER(ctypes.c_double)
KINDS = {"capsule": 0, "halfspace": 1}


class RefFiber(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("material", ctypes.c_int), ("c", ctypes.c_double * 3),
                ("a", ctypes.c_double * 3), ("L", ctypes.c_double), ("R", ctypes.c_double)]


def load(build=True):
    so = os.path.join(_HERE, "c", "libfg_voxel_ref.so")
    if not os.path.exists(so) and build:
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), "libfg_voxel_ref.so"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    lib.ref_voxelize.restype = ctypes.c_int
    lib.ref_voxelize.argtypes = [ctypes.POINTER(RefFiber), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_double, ctypes.c_double, ctypes.c_double, _dp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_double, _dp, _dp, _dp, ctypes.c_char_p, ctypes.c_int]
    return lib


def voxelize(fibers, shape, dims, x0, nphases, matrix_mat, want_normals=False, smooth_levels=-1, smooth_tol=1e-3):
    """Same call as fibergen_amd.geometry.voxelize: (phi[nphases,nx,ny,nz] before normalisation, normals or None,
    {material: real volume fraction}).  `fibers`: objects with kind, material, c, a, L, R."""
    lib = load()
    nx, ny, nz = shape
    arr = (RefFiber * max(len(fibers), 1))()
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
    rc = lib.ref_voxelize(arr, len(fibers), nx, ny, nz, float(dims[0]), float(dims[1]), float(dims[2]),
                          x0a.ctypes.data_as(_dp), nphases, int(matrix_mat), int(smooth_levels), float(smooth_tol),
                          phi.ctypes.data_as(_dp), normals.ctypes.data_as(_dp) if normals is not None else None,
                          real.ctypes.data_as(_dp), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    vol = float(dims[0]) * float(dims[1]) * float(dims[2])
    return phi, normals, {m: real[m] / vol for m in range(nphases)}


def normalize_phi(phi):
    """normalizePhi  F:17613-17626: later materials win, the matrix (whatever is first) keeps the remainder."""
    out = np.array(phi, dtype=np.float64, copy=True)
    rem = np.ones(out.shape[1:])
    for m in range(out.shape[0] - 1, -1, -1):
        out[m] = np.minimum(rem, out[m])
        rem = rem - out[m]
    return out
