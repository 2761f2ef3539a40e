"""Thin object wrapper over the C ABI: one `LSSolver` = one `fg_solver*`.

Mirrors the reference's LSSolver<double,double,3> surface that the project
layer uses (F:14641-24740): materials, phases, normals, options, run(),
means, fields, residuals.  All compute happens in libfibergen_amd.so on the GPU.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib

MIXING = {"voigt": 0, "laminate": 1}

STAGES = {"stress": 0, "div": 1, "fft_forward": 2, "g0": 3, "fft_inverse": 4, "eps": 5, "iteration": 6,
          "stress_const": 7}


# kernels of one pass in launch order (FG_NUM_TIMED_KERNELS slots of fg_get_stage_times)
KERNELS = ["stress", "div", "r2c_z", "c2c_y_fwd", "c2c_x_fwd", "g0", "c2c_x_inv", "c2c_y_inv", "c2r_z", "eps_norm"]


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p)


class LSSolver:
    def __init__(self, nx, ny, nz, dx=1.0, dy=1.0, dz=1.0, device=0):
        self._lib = _lib.load()
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.dx, self.dy, self.dz = float(dx), float(dy), float(dz)
        self._h = self._lib.fg_create(self.nx, self.ny, self.nz, self.dx, self.dy, self.dz, int(device))
        if not self._h:
            raise RuntimeError(self._lib.fg_last_error(None).decode())
        self._cb_keepalive = None
        self.nphases = 0
        self.scalar = False  # heat / porous mode: 3-component gradient / flux, 1-component potential

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.fg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self._lib.fg_last_error(self._h).decode())

    @property
    def shape(self):
        return (self.nx, self.ny, self.nz)

    # -- configuration ----------------------------------------------------
    def set_num_phases(self, n):
        self._check(self._lib.fg_set_num_phases(self._h, int(n)))
        self.nphases = int(n)

    def set_phase(self, p, mu, lam, phi=None):
        ptr = None
        if phi is not None:
            phi = np.ascontiguousarray(phi, dtype=np.float64)
            if phi.shape != self.shape:
                raise ValueError("phi must have shape %r" % (self.shape,))
            ptr = _dp(phi)
        self._check(self._lib.fg_set_phase(self._h, int(p), float(mu), float(lam), ptr))

    def set_normals(self, normals):
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        if normals.shape != (3,) + self.shape:
            raise ValueError("normals must have shape %r" % ((3,) + self.shape,))
        self._check(self._lib.fg_set_normals(self._h, _dp(normals)))

    def set_options(self, **kw):
        for k, v in kw.items():
            if k == "mixing_rule":
                if v not in MIXING:
                    raise RuntimeError("Unknown mixing rule '%s'" % v)
                self._check(self._lib.fg_set_option_i(self._h, b"mixing_rule", MIXING[v]))
            elif k == "method":
                if v not in ("basic", "cg"):
                    raise RuntimeError("Unknown solver method '%s'" % v)
                self._check(self._lib.fg_set_option_i(self._h, b"method", 1 if v == "cg" else 0))
            elif k == "gamma_scheme":
                if v not in ("staggered", "collocated"):
                    raise RuntimeError("Unknown gamma scheme '%s'" % v)
                self._check(self._lib.fg_set_option_i(self._h, b"gamma_scheme", 1 if v == "collocated" else 0))
            elif k == "mode":
                if v not in ("elasticity", "heat", "porous", "viscosity"):
                    raise RuntimeError("mode '%s' is not available on the MI355X path" % v)
                self.scalar = v in ("heat", "porous")
                self._check(self._lib.fg_set_option_i(self._h, b"mode", {"elasticity": 0, "viscosity": 2}.get(v, 1)))
            elif k == "error_estimator":
                kinds = {"epsilon": 0, "residual": 1, "sigma": 2, "energy": 3, "none": 4}   # create_error_estimator  F:14940-14972
                if v not in kinds:
                    raise RuntimeError("Unknown error estimator '%s'" % v)
                self._check(self._lib.fg_set_option_i(self._h, b"error_estimator", kinds[v]))
            elif k in ("u_loop", "fuse_x", "cg_fused", "fuse_stress_div", "u_tile", "x_layout", "plane_fft", "slab_split", "slab_interleave", "slab_loopback", "laminate_overlap", "phi_sweep", "pair_chunk", "joint_x", "tile_plans", "staged_copy", "stage_chunk_kb"):
                self._check(self._lib.fg_set_option_i(self._h, k.encode(), int(v)))
            elif k in ("maxiter", "loadstep_extrapolation_order"):
                self._check(self._lib.fg_set_option_i(self._h, k.encode(), int(v)))
            elif k == "update_ref":
                flag = 0 if v in ("never", 0, False) else 1
                self._check(self._lib.fg_set_option_i(self._h, b"update_ref", flag))
            else:
                self._check(self._lib.fg_set_option_d(self._h, k.encode(), float(v)))

    def set_bc_projector(self, P):
        P = np.ascontiguousarray(P, dtype=np.float64)
        if P.shape != (6, 6):
            raise ValueError("projector must be 6x6")
        self._check(self._lib.fg_set_bc_projector(self._h, _dp(P)))

    def set_convergence_callback(self, fn):
        if fn is None:
            self._cb_keepalive = _lib.CALLBACK()
        else:
            def tramp(_user):
                return 1 if fn() else 0
            self._cb_keepalive = _lib.CALLBACK(tramp)
        self._check(self._lib.fg_set_convergence_callback(self._h, self._cb_keepalive, None))

    def cancel(self):
        self._check(self._lib.fg_cancel(self._h))

    # -- running ------------------------------------------------------------
    def _load6(self, v):
        """prescribed mean value as the 6 doubles the ABI takes (heat / porous: 3 entries, zero padded)"""
        v = np.asarray(v, dtype=np.float64).ravel()
        if self.scalar and v.size == 3:
            v = np.concatenate([v, np.zeros(3)])
        if v.size != 6:
            raise ValueError("prescribed mean value must have %d entries" % (3 if self.scalar else 6))
        return np.ascontiguousarray(v)

    def run(self, E, S=None):
        """LSSolver::run; returns True on error like the reference."""
        E = self._load6(E)
        Sp = None
        if S is not None:
            S = self._load6(S)
            Sp = _dp(S)
        failed = ctypes.c_int(0)
        self._check(self._lib.fg_run_load_case(self._h, _dp(E), Sp, ctypes.byref(failed)))
        return bool(failed.value)

    def run_load_steps(self, E, S=None, params=(0.0, 1.0), first=None, step_callback=None):
        """runLoadsteppingSolver: step i prescribes params[i] * (E, S), continuing from step i-1; step_callback(i) -> True
        stops.  first defaults like the reference (1 for the standard list [0, 1], else 0).  True on error / stop."""
        E = self._load6(E)
        Sp = None
        if S is not None:
            S = self._load6(S)
            Sp = _dp(S)
        par = np.ascontiguousarray(params, dtype=np.float64)
        if first is None:
            first = 0 if par.size > 2 else 1
        cb = _lib.LOADSTEP_CALLBACK(lambda _u, i: 1 if (step_callback is not None and step_callback(i)) else 0)
        failed = ctypes.c_int(0)
        self._check(self._lib.fg_run_load_steps(self._h, _dp(E), Sp, _dp(par), int(par.size), int(first), cb, None,
                                                ctypes.byref(failed)))
        return bool(failed.value)

    def counter(self, name):
        """fg_get_counter: "interface_voxels", "affected_voxels" (lengths of the laminate lists); -1 = unknown name."""
        return int(self._lib.fg_get_counter(self._h, name.encode()))

    def iterate(self, E, n):
        E = self._load6(E)
        self._check(self._lib.fg_iterate(self._h, _dp(E), int(n)))

    def time_iterations(self, E, n):
        """n basic-scheme passes bracketed by HIP events on the solver stream -> milliseconds."""
        E = self._load6(E)
        ms = ctypes.c_double(0.0)
        self._check(self._lib.fg_time_iterations(self._h, _dp(E), int(n), ctypes.byref(ms)))
        return ms.value

    def run_stage(self, stage, E=None):
        sid = STAGES[stage] if isinstance(stage, str) else int(stage)
        Ep = None
        if E is not None:
            E = np.ascontiguousarray(E, dtype=np.float64)
            Ep = _dp(E)
        self._check(self._lib.fg_run_stage(self._h, sid, Ep))

    def synchronize(self):
        self._check(self._lib.fg_synchronize(self._h))

    # -- results --------------------------------------------------------------
    @property
    def iterations(self):
        return int(self._lib.fg_get_iterations(self._h))

    @property
    def residuals(self):
        n = self._lib.fg_get_residuals(self._h, None, 0)
        out = np.zeros(max(n, 1))
        self._lib.fg_get_residuals(self._h, _dp(out), n)
        return out[:n].tolist()

    @property
    def solve_time(self):
        return float(self._lib.fg_get_solve_time(self._h))

    def mean_stress(self):
        out = np.zeros(6)
        self._check(self._lib.fg_mean_stress(self._h, _dp(out)))
        return out[:3] if self.scalar else out

    def mean_strain(self):
        out = np.zeros(6)
        self._check(self._lib.fg_mean_strain(self._h, _dp(out)))
        return out[:3] if self.scalar else out

    def volume_fraction(self, p):
        out = ctypes.c_double(0.0)
        self._check(self._lib.fg_volume_fraction(self._h, int(p), ctypes.byref(out)))
        return out.value

    def calc_ref_material(self):
        self._check(self._lib.fg_calc_ref_material(self._h))
        return self.ref_material

    @property
    def ref_material(self):
        mu, lam = ctypes.c_double(0), ctypes.c_double(0)
        self._lib.fg_get_ref_material(self._h, ctypes.byref(mu), ctypes.byref(lam))
        return mu.value, lam.value

    def get_field(self, name):
        nc = self._lib.fg_field_components(self._h, name.encode())
        if nc <= 0:
            raise RuntimeError("Unknown field '%s'" % name)
        if name == "sumsq":
            out = np.zeros(6)
        elif name == "f_hat":
            nzc = self.nz // 2 + 1
            out = np.empty((3, self.nx, self.ny, nzc, 2))
        else:
            out = np.empty((nc,) + self.shape)   # (fg_get_field writes every element)
        self._check(self._lib.fg_get_field(self._h, name.encode(), _dp(out)))
        if name == "f_hat":
            return out.view(np.complex128)[..., 0]
        return out

    def set_field(self, name, value):
        if name == "f_hat":
            v = np.ascontiguousarray(value, dtype=np.complex128)
            nzc = self.nz // 2 + 1
            if v.shape != (3, self.nx, self.ny, nzc):
                raise ValueError("bad f_hat shape")
            value = v.view(np.float64)
        else:
            value = np.ascontiguousarray(value, dtype=np.float64)
            nc = self._lib.fg_field_components(self._h, name.encode())
            if value.shape != (nc,) + self.shape:
                raise ValueError("field '%s' must have shape %r" % (name, (nc,) + self.shape))
        self._check(self._lib.fg_set_field(self._h, name.encode(), _dp(value)))

    def enable_stage_timing(self, on=True):
        self._check(self._lib.fg_enable_stage_timing(self._h, 1 if on else 0))

    def stage_times(self):
        ms = np.zeros(len(KERNELS))
        cnt = ctypes.c_long(0)
        self._lib.fg_get_stage_times(self._h, _dp(ms), ctypes.byref(cnt))
        return dict(zip(KERNELS, ms.tolist())), cnt.value

    def stage_timing_bias(self):
        """ms an empty HIP-event pair reads on the solver's stream (already subtracted from stage_times)"""
        v = ctypes.c_double(0.0)
        self._lib.fg_get_stage_timing_bias(self._h, ctypes.byref(v))
        return v.value

    def comm_times(self):
        """slab solvers, stage timing on: accumulated ms of the exchanges (fg_get_comm_times)"""
        ms = np.zeros(4)
        self._lib.fg_get_comm_times(self._h, _dp(ms))
        return dict(zip(("alltoall_fwd", "alltoall_bwd", "halo", "allreduce"), ms.tolist()))

    def device_pointer(self, name, comp):
        return self._lib.fg_device_pointer(self._h, name.encode(), int(comp))
