"""`FG` -- the project / action layer of fibergen on top of the MI355X solver.

Mirrors the reference's Python class fibergen.FG (boost.python, F:27142-27187) and the
pieces of FGProject (F:26516-26781), FG<T,R,DIM> (F:24836-26490) and run_actions
(F:25297-26489) that drive the Lippmann-Schwinger hot path:

    fg = FG(); fg.load_xml("project.xml"); fg.set("solver..n", 64); fg.run()
    fg.get_effective_property(); fg.get_field("sigma"); fg.get_mean_stress(); ...

XML subset: <settings> dx,dy,dz,x0,y0,z0, <variables>, <python>, <solver n nx ny nz mult>
with tol, abs_tol, bc_tol, maxiter, method, gamma_scheme, mode, mixing_rule,
error_estimator, ref_scale, update_ref, bc_relax, <laminate_mixing>, <materials>;
actions select_material, place_fiber, print_A2, read_raw_data, init_phase, run_load_case,
calc_effective_properties, calc_isotropic_laminate, python, print_timings, exit,
group-*, skip.  Everything numeric is evaluated as a Python expression over the project
variables, like the reference's embedded interpreter (F:744-756).

All field arithmetic runs in libfibergen_amd.so on the GPU; this module is host glue.
"""
from __future__ import annotations

import builtins
import gzip
import logging
import math
import os

import numpy as np

from . import materials as _materials
from .solver import LSSolver
from .xmlproject import XMLProject

log = logging.getLogger("fibergen_amd")

EXIT_SUCCESS, EXIT_FAILURE = 0, 1
_VOIGT = (11, 22, 33, 23, 13, 12)
_SOLVER_DOUBLE_KEYS = ("tol", "abs_tol", "bc_tol", "ref_scale", "bc_relax")


class _Fiber:
    def __init__(self, kind, c, a, L, R, material):
        self.kind, self.c, self.a, self.L, self.R, self.material = kind, np.array(c, float), np.array(a, float), L, R, material


class FG:
    """The fibergen solver class (reference: PyFG, F:26785-26851)."""

    def __init__(self, device=0):
        self._device = device
        self._shard = False
        self._shard_group = None
        self._slabs = False
        self._slab_group = None
        self._project = XMLProject()
        self._variables = {}
        self._py_enabled = True
        self._xml_precision = -1
        self._convergence_callback = None
        self._loadstep_callback = None
        self._log_file = None
        self._Ceff_voigt = None
        self._timings = {}
        self._injected_phi = {}      # extension: phase arrays handed in by the caller, survive run()
        self._injected_normals = None
        self._reset_state()

    # ------------------------------------------------------------------ state
    def _reset_state(self):
        """FG::reset  F:24932-24942"""
        lss = getattr(self, "_lss", None)
        if lss is not None:
            lss.close()
        self._lss = None
        self._error = None
        self._fibers = []
        self._selected_material = 0
        self._phase_valid = False
        self._solver_valid = False
        self._raw_phase = False
        self._phase_names = []
        self._phase_materials = []
        self._matrix_mat = 0
        self._phi = None        # host copy of the phase fields [nphase][nx][ny][nz]
        self._normals = None
        self._raw_normals = None
        self._want_normals = False
        self._method = "cg"
        self._mode = "elasticity"
        self._real_vf = {}

    def reset(self):
        """Resets the solver to its initial state and unloads any loaded XML (F:26536-26541)."""
        self._project.reset()
        self._injected_phi = {}
        self._injected_normals = None
        self._reset_state()

    # ------------------------------------------------------------------ XML access
    def load_xml(self, filename):
        self._project.load_xml(filename)

    def set_xml(self, xml):
        self._project.set_xml(xml)

    def get_xml(self):
        return self._project.get_xml()

    def set_xml_precision(self, digits):
        self._xml_precision = int(digits)

    def get_xml_precision(self):
        return self._xml_precision

    def _format_value(self, value):
        if isinstance(value, bool):
            return "1" if value else "0"
        if isinstance(value, int):
            return str(value)
        if isinstance(value, float):
            if self._xml_precision >= 0:
                return "%.*g" % (self._xml_precision, value)
            return repr(value)
        if isinstance(value, str):
            return value
        raise RuntimeError("invalid argument for attribute specified")

    def set(self, path, *args, **kwargs):
        """set(path), set(path, value), set(path, a=1, b=2)  (SetParameters, F:26854-26901)"""
        if not args and not kwargs:
            self._project.set(path, "")
        for v in args:
            self._project.set(path, self._format_value(v))
        for k, v in kwargs.items():
            try:
                self._project.set(path + "." + k, self._format_value(v))
            except RuntimeError:
                raise RuntimeError("invalid argument for attribute '%s' specified" % (path + "." + k))

    def get(self, path):
        return self._project.get(path)

    def erase(self, path):
        self._project.erase(path)

    # ------------------------------------------------------------------ python evaluation
    def set_py_enabled(self, enabled):
        self._py_enabled = bool(enabled)

    def set_variable(self, name, value):
        self._variables[name] = value

    def get_variable(self, name):
        return self._variables[name]

    def _eval(self, text, typ=float):
        """PY::eval<T>  F:744-756: Python expression over the project variables when enabled."""
        if isinstance(text, (int, float)):
            return typ(text)
        s = str(text).strip()
        if self._py_enabled:
            try:
                v = eval(s, {"__builtins__": builtins, "math": math, "np": np}, self._variables)
            except Exception as e:
                raise RuntimeError("error evaluating expression '%s': %s" % (s, e))
            return typ(v)
        if typ is bool:
            return s not in ("0", "false", "False", "")
        if typ is int:
            return int(float(s))
        return typ(s)

    def _attr(self, el, name, default=None, typ=float):
        if el is None or name not in el.attrib:
            if default is None and typ is not str:
                raise RuntimeError("Undefined property: %s!" % name)
            return default
        if typ is str:
            return el.attrib[name]
        return self._eval(el.attrib[name], typ)

    def _child_value(self, el, name, default, typ=float):
        c = el.find(name) if el is not None else None
        if c is None or c.text is None or c.text.strip() == "":
            return default
        if typ is str:
            return c.text.strip()
        return self._eval(c.text, typ)

    # ------------------------------------------------------------------ callbacks / misc API
    def set_convergence_callback(self, func):
        self._convergence_callback = func

    def shard_load_cases(self, enable=True, group=None):
        """Extension of the reference API for multi-GPU jobs (one process per GPU, torch.distributed initialised, this FG
        created with device = LOCAL_RANK): calc_effective_properties gives each rank every world_size-th of its six unit
        experiments and gathers the mean stresses -- the load cases are independent, no field ever crosses a link.  Every
        rank ends with the same effective stiffness.  (The slab decomposition of ONE load case is
        fibergen_amd.distributed.DistributedLSSolver.)"""
        self._shard = bool(enable)
        self._shard_group = group

    def decompose_slabs(self, enable=True, group=None):
        """Extension of the reference API for multi-GPU jobs (one process per GPU, torch.distributed initialised, this FG
        created with device = LOCAL_RANK): the voxel grid of every load case is cut into x-slabs over the ranks
        (fibergen_amd.distributed.DistributedLSSolver: RCCL all-to-all between the FFT axes, halo planes, all-reduced
        norms).  Every rank runs the same project and sees the same results; fields returned by get_field are gathered.
        nx and ny must be divisible by the number of ranks; <loadsteps> without extrapolation; elasticity, viscosity, heat and
        porous (Voigt mixing; CG with prescribed mean gradients)."""
        self._slabs = bool(enable)
        self._slab_group = group
        self._solver_valid = False
        self._phase_valid = False

    def _load_case_shard(self):
        if not getattr(self, "_shard", False):
            return 0, 1
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0, 1
        return dist.get_rank(self._shard_group), dist.get_world_size(self._shard_group)

    def set_loadstep_callback(self, func):
        self._loadstep_callback = func

    def set_log_file(self, filename):
        self._log_file = filename
        handler = logging.FileHandler(filename)
        log.addHandler(handler)
        log.setLevel(logging.INFO)

    def cancel(self):
        """FG::cancel  F:25190-25193"""
        self._error = "fibergen canceled"
        if self._lss is not None:
            self._lss.cancel()

    def get_error(self):
        return self._error is not None

    # ------------------------------------------------------------------ initialisation
    def _settings(self):
        return self._project.root

    def _grid(self):
        st = self._settings()
        solver = st.find("solver")
        na = self._attr(solver, "n", 0, int)
        mult = self._attr(solver, "mult", 1.0, float)
        dims = []
        for key in ("nx", "ny", "nz"):
            v = self._attr(solver, key, na, int)
            dims.append(max(1, int(v * mult)))
        d = [self._child_value(st, k, 1.0) for k in ("dx", "dy", "dz")]
        x0 = [self._child_value(st, k, 0.0) for k in ("x0", "y0", "z0")]
        return tuple(dims), d, x0

    def init_lss(self):
        """FG::init_lss + LSSolver::readSettings  F:24990-25017, F:15044-15363"""
        if self._solver_valid:
            return
        st = self._settings()
        solver = st.find("solver")
        (nx, ny, nz), (dx, dy, dz), x0 = self._grid()
        self._dims = (dx, dy, dz)
        self._x0 = x0
        mode = self._child_value(solver, "mode", "elasticity", str)
        if mode not in ("elasticity", "heat", "porous", "viscosity"):
            raise RuntimeError("mode '%s' is not available on the MI355X path (elasticity, heat, porous, viscosity)" % mode)
        self._mode = mode
        scalar = mode != "elasticity"      # laws with the single constant mu (ScalarLinearIsotropicMaterialLaw)
        method = self._child_value(solver, "method", "cg", str)
        if method not in ("basic", "cg"):
            raise RuntimeError("Unknown solver method '%s'" % method)
        self._method = method
        scheme = self._child_value(solver, "gamma_scheme", "auto", str)
        if scheme == "auto":
            scheme = "staggered"
        if scheme not in ("staggered", "collocated") or (scheme == "collocated" and scalar):  # scalar: any non-elastic mode
            raise RuntimeError("gamma scheme '%s' is not built on the MI355X path (staggered; collocated for elasticity): set "
                               "<gamma_scheme>staggered</gamma_scheme> -- the schemes differ by their discretisation error, "
                               "nothing is substituted silently" % scheme)
        est = self._child_value(solver, "error_estimator", "epsilon", str)
        if est not in ("epsilon", "residual", "sigma", "energy", "none"):   # create_error_estimator  F:14940-14972
            if est == "div_sigma":   # DivSigmaErrorEstimator F:14473-14510 is a stub in the reference (abs = rel = 0: one iteration)
                raise RuntimeError("error estimator 'div_sigma' is not available (a stub in the reference: it stops after one iteration)")
            raise RuntimeError("Unknown error estimator '%s'" % est)
        mixing = self._child_value(solver, "mixing_rule", "voigt", str)
        if mixing not in ("voigt", "laminate"):
            raise RuntimeError("Unknown mixing rule '%s'" % mixing)
        if scalar and mixing != "voigt":
            raise RuntimeError("mixing rule '%s' is not available in %s mode (voigt only)" % (mixing, mode))

        if getattr(self, "_slabs", False):
            from .distributed import GlobalViewSolver
            if scheme != "staggered":
                raise RuntimeError("slab decomposition runs the staggered Green operator")
            lss = GlobalViewSolver(nx, ny, nz, dx, dy, dz, group=getattr(self, "_slab_group", None), device=self._device)
        else:
            lss = LSSolver(nx, ny, nz, dx, dy, dz, device=self._device)
        opts = {"mode": mode, "mixing_rule": mixing, "method": method, "gamma_scheme": scheme, "error_estimator": est}
        # <loadsteps>  F:15095-15119: a count (uniform steps i / n) or a list of <loadstep param=".."/>
        self._loadsteps = [0.0, 1.0]
        ls = solver.find("loadsteps") if solver is not None else None
        if ls is not None:
            entries = [c for c in ls if isinstance(c.tag, str)]
            if entries:
                self._loadsteps = [self._eval(c.attrib["param"]) for c in entries if c.tag == "loadstep"]
            else:
                n = self._eval(ls.text.strip(), int) if (ls.text and ls.text.strip()) else 1
                self._loadsteps = [i / float(n) for i in range(n + 1)]
        self._first_loadstep = self._child_value(solver, "first_loadstep", -1, int)
        self._write_loadsteps = bool(self._child_value(solver, "write_loadsteps", 0, int))
        self._loadstep_filename = self._child_value(solver, "loadstep_filename", "loadstep_%02d.vtk", str)
        # F:15090-15091; "transformation" (F:21516-21580) belongs to hyperelasticity, which is out of scope
        self._extrapolation_order = self._child_value(solver, "loadstep_extrapolation_order", 0, int)
        if self._child_value(solver, "loadstep_extrapolation_method", "polynomial", str) != "polynomial" and self._extrapolation_order > 0:
            raise RuntimeError("loadstep_extrapolation_method: only 'polynomial' is available on the MI355X path")
        opts["loadstep_extrapolation_order"] = self._extrapolation_order
        for k in _SOLVER_DOUBLE_KEYS:
            v = self._child_value(solver, k, None)
            if v is not None:
                opts[k] = v
        v = self._child_value(solver, "maxiter", None, int)
        if v is not None:
            opts["maxiter"] = v
        upd = self._child_value(solver, "update_ref", None, str)
        if upd is not None:
            opts["update_ref"] = upd
        lam = solver.find("laminate_mixing") if solver is not None else None
        if lam is not None:
            for k in ("eps_g", "eps_a"):
                v = self._child_value(lam, k, None)
                if v is not None:
                    opts[k] = v
        # <materials>  F:15177-15299
        mats = solver.find("materials") if solver is not None else None
        names, consts = [], []
        matrix_mat, matrix_set = 0, False
        if mats is not None:
            for m in mats:
                if not isinstance(m.tag, str):
                    continue
                attrs = dict(m.attrib)
                if m.tag == "ref":
                    if scalar:
                        opts["mu_0"] = self._eval(attrs.get("mu", "1"))
                        continue
                    c = _materials.material_constants(attrs, self._eval)
                    opts["mu_0"], opts["lambda_0"] = c["mu"], c["lambda"]
                    continue
                if m.tag == "matrix" or self._eval(attrs.get("matrix", "0"), int) != 0:
                    if matrix_set:
                        raise RuntimeError("Matrix material already specified")
                    matrix_set = True
                    matrix_mat = len(names)
                law = attrs.get("law", "iso")
                if law != "iso":
                    raise RuntimeError("Unknown material law '%s'" % law)
                names.append(m.tag)
                if scalar:
                    # ScalarLinearIsotropicMaterialLaw::readSettings  F:11170-11173: one constant, mu (default 1)
                    consts.append({"mu": self._eval(attrs.get("mu", "1")), "lambda": 0.0})
                else:
                    consts.append(_materials.material_constants(attrs, self._eval))
        if not names:
            raise RuntimeError("No materials specified")
        lss.set_num_phases(len(names))
        for p, c in enumerate(consts):
            lss.set_phase(p, c["mu"], c["lambda"])
        lss.set_options(**opts)
        lss.set_convergence_callback(self._on_iteration)
        self._lss = lss
        self._phase_names = names
        self._phase_materials = consts
        self._matrix_mat = matrix_mat
        self._mixing = mixing
        self._solver_valid = True

    def init_fibers(self):
        """Random fibre generation is out of scope; placed fibres need no generation."""
        return

    def _material_id(self, name):
        try:
            return self._phase_names.index(name)
        except ValueError:
            raise RuntimeError("Material '%s' not found" % name)

    def init_phase(self):
        """FG::init_phase  F:25026-25038: discretise the geometry into phase fractions (+ normals)."""
        if self._phase_valid:
            return
        self.init_lss()
        lss = self._lss
        shape = lss.shape
        nph = len(self._phase_names)
        for name, arr in self._injected_phi.items():
            m = self._material_id(name)
            if arr.shape != shape:
                raise RuntimeError("Phase dimensions are incompatible")
            if not self._raw_phase:
                self._phi = np.ones((nph,) + shape)  # setPhasesOne  F:17128
                self._raw_phase = True
            self._phi[m] = arr
        if self._injected_normals is not None:
            if self._injected_normals.shape != (3,) + shape:
                raise RuntimeError("normals must have shape (3, nx, ny, nz)")
            self._normals = self._injected_normals
        elif self._raw_phase and getattr(self, "_raw_normals", None) is not None and \
                (self._want_normals or self._mixing == "laminate"):
            self._normals = self._raw_normals   # initMultiphase centroid normals  F:16861-16907
        if self._raw_phase:
            phi = self._phi
        else:
            from . import geometry
            phi, normals, real_vf = geometry.voxelize(self._fibers, shape, self._dims, self._x0, nph, self._matrix_mat,
                                                      want_normals=(self._want_normals or self._mixing == "laminate"),
                                                      smooth_levels=self._solver_int("smooth_levels", -1),
                                                      smooth_tol=self._solver_float("smooth_tol", 0.001),
                                                      device=self._device)
            if normals is not None:
                self._normals = normals
            self._real_vf = real_vf
        phi = _normalize_phi(phi)  # normalizePhi  F:17588-17646 (last material wins)
        self._phi = phi
        for p in range(nph):
            lss.set_phase(p, self._phase_materials[p]["mu"], self._phase_materials[p]["lambda"], phi[p])
        if self._normals is not None:
            lss.set_normals(self._normals)
        elif self._mixing == "laminate":
            raise RuntimeError("laminate mixing needs interface normals (init_phase normals=\"1\" or set_normals)")
        self._phase_valid = True

    def _solver_float(self, key, default):
        return self._child_value(self._settings().find("solver"), key, default)

    def _solver_int(self, key, default):
        return self._child_value(self._settings().find("solver"), key, default, int)

    # extension of the reference API: inject phase fractions / normals as arrays
    def set_phase_field(self, name, phi):
        """Provide the volume-fraction field of material `name` as an array [nx,ny,nz]
        (equivalent to read_raw_data with dtype=double, order=row).  Unlike the XML tree this
        is not part of the project file; it stays in effect for later run() calls until
        reset()."""
        self._injected_phi[name] = np.clip(np.asarray(phi, dtype=np.float64), 0.0, 1.0)
        self._phase_valid = False

    def set_normals(self, normals):
        """Provide the interface normals [3,nx,ny,nz] (see set_phase_field)."""
        self._injected_normals = np.asarray(normals, dtype=np.float64)
        self._phase_valid = False

    # ------------------------------------------------------------------ results
    def get_phase_names(self):
        self.init_lss()
        return list(self._phase_names)

    def get_volume_fraction(self, name):
        self.init_lss()
        return self._lss.volume_fraction(self._material_id(name))

    def get_real_volume_fraction(self, name):
        self.init_lss()
        return float(self._real_vf.get(self._material_id(name), float("nan")))

    def get_residuals(self):
        self.init_lss()
        return self._lss.residuals

    def get_solve_time(self):
        self.init_lss()
        return self._lss.solve_time

    def get_mean_stress(self):
        self.init_lss()
        return self._lss.mean_stress().tolist()

    def get_mean_strain(self):
        self.init_lss()
        return self._lss.mean_strain().tolist()

    def get_mean_energy(self):
        """calcMeanEnergy (F:17784): <W> = 1/2 <sigma : epsilon> for linear phases."""
        self.init_lss()
        s = self._lss.get_field("sigma")
        e = self._lss.get_field("epsilon")
        if self._mode != "elasticity":
            return float(0.5 * (s * e).sum(axis=0).mean())
        w = (s[:3] * e[:3]).sum(axis=0) + 2 * (s[3:] * e[3:]).sum(axis=0)
        return float(0.5 * w.mean())

    def get_effective_property(self):
        if self._Ceff_voigt is None:
            return []
        return [list(map(float, row)) for row in self._Ceff_voigt]

    def get_rve_dims(self):
        (_, _, _), (dx, dy, dz), x0 = self._grid()
        return [x0[0], x0[1], x0[2], dx, dy, dz]

    def get_distance_evals(self):
        return 0

    def _axes(self):
        """unit axes of the fibres placed so far (place_fiber; the random generator is outside the MI355X path)"""
        if not self._fibers:
            raise RuntimeError("no fibres placed: orientation moments are undefined")
        a = np.array([f.a for f in self._fibers], dtype=float)
        return a / np.sqrt((a * a).sum(axis=1))[:, None]

    def get_A2(self):
        """FiberGenerator::updateMoments F:6263-6275 + getA2 F:6683-6686: sum of a (x) a over the fibres, trace-normalised"""
        a = self._axes()
        A2 = np.einsum("ni,nj->ij", a, a)
        return (A2 / np.trace(A2)).tolist()

    def get_A4(self):
        """getA4 F:6689-6707: sum of a (x) a (x) a (x) a, scaled by the trace of its contraction A4_iikl"""
        a = self._axes()
        A4 = np.einsum("ni,nj,nk,nl->ijkl", a, a, a, a)
        return (A4 / np.trace(np.einsum("iikl->kl", A4))).tolist()

    def get_B_from_A(self, a0, a1, a2):
        raise RuntimeError("fibre orientation statistics are outside the MI355X hot path")

    def get_field(self, name, range_x=(), range_y=(), range_z=(), components=()):
        """GetField  F:26931-27010: ndarray[ncomp, nx, ny, nz] (float64), optionally sub-sampled."""
        self.init_lss()
        lss = self._lss
        if name in ("epsilon", "sigma", "u"):
            data = lss.get_field(name)
        elif name == "phi":
            self.init_phase()
            data = lss.get_field("phi")
        elif name == "normals":
            self.init_phase()
            data = lss.get_field("normals")
        elif name in self._phase_names:
            self.init_phase()
            data = lss.get_field("phi")[self._material_id(name)][None]
        else:
            raise RuntimeError("Unknown field '%s'" % name)

        def rng(r, n):
            if len(r) == 0:
                return np.arange(n)
            r = np.unique(np.asarray(r, dtype=np.int64))
            if r.min() < 0 or r.max() >= n:
                raise IndexError("index out of range")
            return r
        ic = rng(components, data.shape[0])
        ix, iy, iz = rng(range_x, data.shape[1]), rng(range_y, data.shape[2]), rng(range_z, data.shape[3])
        return np.ascontiguousarray(data[np.ix_(ic, ix, iy, iz)])

    def _run_lss(self, E, S=None):
        """LSSolver::run with the project's load steps (runLoadsteppingSolver F:21584-21685); after every step the
        load-step actions (performLoadstepActions F:21435-21447): VTK file if <write_loadsteps>, then the callback.
        Returns the reference's run() value (True = error or stop request)."""
        def step(istep):
            if self._write_loadsteps and self._loadstep_filename:
                self.write_vtk(self._loadstep_filename % istep)
            return self._loadstep_callback is not None and bool(self._loadstep_callback())
        first = self._first_loadstep if self._first_loadstep >= 0 else (0 if len(self._loadsteps) > 2 else 1)
        return self._lss.run_load_steps(E, S, self._loadsteps, first, step)

    def get_mean_cauchy_stress(self):
        """FG::get_mean_cauchy_stress  F:27177 / F:25122: for the small-strain laws the Cauchy stress is the stress"""
        return self.get_mean_stress()

    def _effective_viscosity(self, outdir):
        """calc_effective_properties, viscosity branch  F:26252-26347: five traceless stress experiments,
        Ceff55 = E55 S55^-1, completion to 6x6 by tracelessness, row shift of the first three columns,
        Voigt halving of the last three."""
        E = np.zeros((6, 5))
        E[0, 0] = E[1, 1] = 1
        E[1, 0] = E[2, 1] = -1
        E[3, 2] = E[4, 3] = E[5, 4] = 1
        S = np.zeros((6, 5))
        for i in range(5):
            failed = self._run_lss(E[:, i])
            if failed or self._error is not None:
                self._error = self._error or "NaN detected in solution. Aborting."
                return EXIT_FAILURE
            S[:, i] = self._lss.mean_stress()
            if outdir:
                self.write_vtk("%s/results_%d.vtk" % (outdir, i + 1))
        E55, S55 = E[1:6, :], S[1:6, :]
        try:
            C55 = E55 @ np.linalg.inv(S55)
        except np.linalg.LinAlgError:
            C55 = np.eye(5) * np.inf
        C = np.zeros((6, 6))
        C[1:6, 1:6] = C55
        for i in range(5):
            if S[0, i] != 0:
                for j in range(1, 6):
                    C[j, 0] = (E[j, i] - C[j, 1:6] @ S[1:6, i]) / S[0, i]
                break
        C[0, :] = -(C[1, :] + C[2, :])
        C[:, :3] -= C[:, :3].min(axis=1)[:, None]
        Cv = C.copy()
        Cv[:, 3:6] *= 0.5
        self._Ceff_voigt = Cv
        log.info("Effective viscosity matrix \"2*eta\" (Voigt notation):\n%s", Cv)
        return None

    def write_vtk(self, filename):
        """LSSolver::writeVTK  F:23317-23451: phase fractions, strain, stress and displacement of the current
        state as a legacy VTK file (format from <res_format>, value type from <restype>, F:25300, F:26552)."""
        from . import vtk
        self.init_lss()
        self.init_phase()
        st = self._settings()
        lss = self._lss
        eps, sig = lss.get_field("epsilon"), lss.get_field("sigma")
        if self._mode == "viscosity":
            # dual scheme: the solver's "epsilon" is the fluid stress, its "sigma" the shear rate (F:23403-23416);
            # the pressure field of the reference's file (a Poisson solve, F:23418-23430) is not written
            eps, sig = sig, eps
        vtk.write_results(filename, lss.shape, self._dims, self._x0, self._phase_names, lss.get_field("phi"),
                          eps, sig, lss.get_field("u"),
                          binary=self._child_value(st, "res_format", "binary", str) == "binary",
                          dtype=self._child_value(st, "restype", "float", str),
                          mode="elasticity" if self._mode == "viscosity" else self._mode)

    # ------------------------------------------------------------------ running
    def _on_iteration(self):
        if self._error is not None:
            return True
        res = self._lss.residuals
        log.info("# Iteration %d: epsilon error rel. = %g", len(res), res[-1] if res else float("nan"))
        if self._convergence_callback is not None:
            r = self._convergence_callback()
            if isinstance(r, (bool, np.bool_)) and r:
                log.info("Custom convergence test returned true.")
                return True
        return False

    def _init_python(self):
        """FG::init_python  F:24873-24930: <variables> then <python> blocks."""
        self._variables["fg"] = self
        st = self._settings()
        variables = st.find("variables")
        if variables is not None:
            for v in variables:
                if not isinstance(v.tag, str):
                    continue
                typ = v.attrib.get("type", "object")
                val = v.attrib.get("value", "")
                if typ == "str":
                    pv = val
                elif typ == "int":
                    pv = self._eval(val, int)
                elif typ == "float":
                    pv = self._eval(val, float)
                elif typ == "object":
                    pv = eval(val.strip(), {"__builtins__": builtins, "math": math, "np": np}, self._variables)
                else:
                    raise RuntimeError("Unknown variable type '%s' for %s" % (typ, v.tag))
                self._variables[v.tag] = pv
        for p in st.findall("python"):
            self._exec_python(p.text or "")

    def _exec_python(self, code):
        import textwrap
        g = {"__builtins__": builtins}
        g.update(self._variables)
        exec(textwrap.dedent(code), g)
        for k, v in g.items():
            if k != "__builtins__":
                self._variables[k] = v

    def run(self, path="actions"):
        """FG::run  F:25195-25295: reset, evaluate variables, perform the actions below `path`."""
        self._reset_state()
        self._init_python()
        try:
            return self._run_actions(path)
        finally:
            self._variables.pop("fg", None)

    def _run_actions(self, path):
        st = self._settings()
        node = st
        for part in path.split("."):
            node = node.find(part) if node is not None else None
        if node is None:
            return EXIT_SUCCESS
        if self._eval(node.attrib.get("skip", "0"), int) != 0:
            return EXIT_SUCCESS
        for act in node:
            if self._error is not None:
                return EXIT_FAILURE
            if not isinstance(act.tag, str):
                continue  # comment
            if act.tag == "skip" or self._eval(act.attrib.get("skip", "0"), int) != 0:
                continue
            if act.tag.startswith("group-"):
                ret = self._run_actions(path + "." + act.tag)
                if ret != 0:
                    return ret
                continue
            ret = self._run_action(act)
            if ret == "exit":
                return EXIT_SUCCESS
            if ret not in (None, 0):
                return ret
        return EXIT_FAILURE if self._error is not None else EXIT_SUCCESS

    def _voigt_vector(self, act, prefix):
        """read_voigt_vector  F:1126-1138: e1..e3 shorthand, then e11,e22,e33,e23,e13,e12"""
        v = np.zeros(6)
        for i in range(3):
            v[i] = self._attr(act, "%s%d" % (prefix, i + 1), v[i])
        for i, idx in enumerate(_VOIGT):
            v[i] = self._attr(act, "%s%d" % (prefix, idx), v[i])
        return v

    def _run_action(self, act):
        name = act.tag
        if name == "select_material":
            self.init_lss()
            self._selected_material = self._material_id(self._attr(act, "name", None, str) or "")
            return None
        if name == "place_fiber":
            (_, _, _), (dx, dy, dz), x0 = self._grid()
            L = self._attr(act, "L", 0.0)
            R = self._attr(act, "R", 0.25 * dx)
            V = self._attr(act, "V", -1.0)
            kind = self._attr(act, "type", "capsule", str)
            if V >= 0:
                R = (V / (4 * math.pi / 3.0)) ** (1 / 3.0)
            c = [self._attr(act, "cx", x0[0] + 0.5 * dx), self._attr(act, "cy", x0[1] + 0.5 * dy),
                 self._attr(act, "cz", x0[2] + 0.5 * dz)]
            a = [self._attr(act, "ax", 1.0), self._attr(act, "ay", 0.0), self._attr(act, "az", 0.0)]
            if kind not in ("capsule", "halfspace"):
                raise RuntimeError("Unknown fiber type '%s'" % kind)
            self._fibers.append(_Fiber(kind, c, a, L, R, self._selected_material))
            self._phase_valid = False
            return None
        if name == "read_raw_data":
            return self._read_raw_data(act)
        if name == "init_phase":
            self.init_lss()
            if self._attr(act, "normals", False, bool):
                self._want_normals = True
            self.init_phase()
            return None
        if name == "run_load_case":
            self.init_lss()
            scalar = self._mode in ("heat", "porous")
            E = self._voigt_vector(act, "e")
            S = self._voigt_vector(act, "s")
            P = np.diag([1.0, 1.0, 1.0, 0.5, 0.5, 0.5])
            dim = 3 if scalar else 6
            for i in range(dim):
                for j in range(dim):
                    key = "p%d%d" % (i + 1, j + 1)
                    if key in act.attrib:
                        P[i, j] = P[j, i] = self._eval(act.attrib[key])
            if scalar:
                # heat / porous branch  F:26002-26024: 3-vectors e1..e3 / s1..s3 (or e11, ...), 3x3 projector p11..p33
                # (Voigt::Id4(3) = Id); handed to the 6x6 interface with an inert shear block
                P3 = P[:3, :3].copy()
                P = np.diag([1.0, 1.0, 1.0, 0.5, 0.5, 0.5])
                P[:3, :3] = P3
                E, S = E[:3], (S[:3] if np.abs(S[:3]).max() > 0 or np.abs(P3 - np.eye(3)).max() > 0 else None)
            if self._mode == "viscosity":
                # F:25975-25989: prescribed fluid stress and shear rate must be traceless
                tol = 100.0 * np.finfo(np.float64).eps
                if abs(E[0] + E[1] + E[2]) > tol:
                    raise RuntimeError("Prescibed fluid stress %s has not zero trace!" % E)
                if abs(S[0] + S[1] + S[2]) > tol:
                    raise RuntimeError("Prescibed fluid strain %s has not zero trace!" % S)
            self.init_phase()
            self._lss.set_bc_projector(P)
            stopped = [False]
            user_cb = self._loadstep_callback
            if user_cb is not None:   # a stop request ends the run, but run()'s return value is ignored here (F:25938)
                def wrapped():
                    stopped[0] = bool(user_cb())
                    return stopped[0]
                self._loadstep_callback = wrapped
            try:
                failed = self._run_lss(E, S)
            finally:
                self._loadstep_callback = user_cb
            if failed and not stopped[0]:
                self._error = self._error or "NaN detected in solution. Aborting."
                return EXIT_FAILURE
            if self._error is not None:
                return EXIT_FAILURE
            outfile = self._attr(act, "outfile", "", str)
            if outfile:
                self.write_vtk(outfile)
            return None
        if name in ("write_lss_vtk", "write_vtk2"):
            # FG::run_actions  F:25374-25380 (filename=) and F:25427-25436 (outfile=)
            fn = self._attr(act, "filename" if name == "write_lss_vtk" else "outfile", None, str)
            if not fn:
                raise RuntimeError("%s needs a file name" % name)
            self.write_vtk(fn)
            return None
        if name == "write_raw_data":
            # FG::run_actions  F:25448-25493 + writeRawPhase  F:17004-17074: phase fraction * scale cast to dtype
            # (C truncation), column order = x fastest in the file (default) or row order = z fastest; .gz compressed
            fn = self._attr(act, "filename", None, str)
            dtype = self._attr(act, "dtype", "uint8", str)
            col = self._attr(act, "order", "col", str) == "col"
            self.init_lss()
            m = self._material_id(self._attr(act, "material", "", str))
            self.init_phase()
            types = {"uint8": (np.uint8, 0.9999 + 0xff), "uint16": (np.uint16, 0.9999 + 0xffff),
                     "uint32": (np.uint32, 0.9999 + 0xffffffff), "float": (np.float32, 1.0), "double": (np.float64, 1.0)}
            if dtype not in types:
                raise RuntimeError("Unknown dtype '%s'" % dtype)
            typ, scale = types[dtype]
            scale = self._attr(act, "scale", scale)
            phi = self._lss.get_field("phi")[m] * scale
            data = phi.astype(typ)  # float -> integer conversion truncates like the C cast
            raw = (data.transpose(2, 1, 0) if col else data).tobytes()
            if fn.endswith(".gz"):
                import gzip
                with gzip.open(fn, "wb") as f:
                    f.write(raw)
            else:
                with open(fn, "wb") as f:
                    f.write(raw)
            return None
        if name == "write_vtk_phase":
            self.init_lss()
            self.init_phase()
            from . import vtk
            m = self._material_id(self._attr(act, "name", "", str))
            st = self._settings()
            vtk.write_phase(self._attr(act, "outfile", None, str), self._lss.shape, self._dims, self._x0,
                            self._phase_names[m], self._lss.get_field("phi")[m],
                            binary=self._child_value(st, "res_format", "binary", str) == "binary",
                            dtype=self._child_value(st, "restype", "float", str))
            return None
        if name == "calc_effective_properties":
            self.init_lss()
            self.init_phase()
            outdir = self._attr(act, "outdir", "", str)
            if self._mode == "viscosity":
                return self._effective_viscosity(outdir)
            if self._mode != "elasticity":
                # heat / porous branch  F:26115-26165: three unit gradients, Ceff = S E^-1 (3x3)
                S = np.zeros((3, 3))
                for i in range(3):
                    Ep = np.zeros(3)
                    Ep[i] = 1.0
                    failed = self._run_lss(Ep)
                    if failed or self._error is not None:
                        self._error = self._error or "NaN detected in solution. Aborting."
                        return EXIT_FAILURE
                    S[:, i] = self._lss.mean_stress()
                    if outdir:
                        self.write_vtk("%s/results_%d.vtk" % (outdir, i + 1))
                self._Ceff_voigt = S @ np.linalg.inv(np.eye(3))
                log.info("Effective %s matrix:\n%s", "conductivity" if self._mode == "heat" else "permeability",
                         self._Ceff_voigt)
                return None
            S = np.zeros((6, 6))
            rank, world = self._load_case_shard()
            bad = False
            for i in range(rank, 6, world):   # the six unit experiments are independent: one rank each (no data-path collective)
                Ep = np.zeros(6)
                Ep[i] = 1.0
                failed = self._run_lss(Ep)
                if failed or self._error is not None:
                    bad = True
                    break
                S[:, i] = self._lss.mean_stress()
                if outdir:
                    self.write_vtk("%s/results_%d.vtk" % (outdir, i + 1))  # F:26056-26062
            if world > 1:
                import torch.distributed as dist
                parts = [None] * world
                dist.all_gather_object(parts, (bad, S), group=self._shard_group)
                bad = any(b for b, _ in parts)
                S = sum(Sp for _, Sp in parts)
            if bad:
                self._error = self._error or "NaN detected in solution. Aborting."
                return EXIT_FAILURE
            Ceff = S @ np.linalg.inv(np.eye(6))  # Ceff = S E^-1 with unit experiments  F:26072-26075
            Cv = Ceff.copy()
            Cv[:, 3:6] *= 0.5                     # F:26083-26088
            self._Ceff_voigt = Cv
            log.info("Effective stiffness matrix (Voigt notation):\n%s", Cv)
            return None
        if name == "calc_isotropic_laminate":
            c = np.zeros(6)
            for m in act:
                if not isinstance(m.tag, str):
                    continue
                k = _materials.material_constants(dict(m.attrib), self._eval)
                phi = self._eval(m.attrib.get("phi", "0"))
                lam, mu = k["lambda"], k["mu"]
                c += phi * np.array([1 / (lam + 2 * mu), 1 / mu, mu, lam / (lam + 2 * mu),
                                     4 * mu * (lam + mu) / (lam + 2 * mu), 2 * mu * lam / (lam + 2 * mu)])
            C = np.zeros((6, 6))
            C[0, 0] = 1 / c[0]
            C[1, 1] = C[2, 2] = c[4] + c[3] * c[3] / c[0]
            C[3, 3] = c[2]
            C[4, 4] = C[5, 5] = 1 / c[1]
            C[0, 1] = C[1, 0] = C[0, 2] = C[2, 0] = c[3] / c[0]
            C[1, 2] = C[2, 1] = c[5] + c[3] * c[3] / c[0]
            self._laminate_Ceff = C
            log.info("Effective stiffness matrix (Voigt notation):\n%s", C)
            return None
        if name == "calc_HS_bounds":
            # Hashin-Shtrikman bounds of a two-phase isotropic mixture  F:25730-25742, HashinBounds::get F:7463-7484; the two
            # materials are read with the postfixes "1" / "2" (Material("", "1").readSettings: mu1 + lambda1, E1 + nu1, ..., phi1)
            ms = []
            for post in ("1", "2"):
                attrs = {k[:-1]: v for k, v in act.attrib.items() if k.endswith(post) and k[:-1] in _materials.NAMES}
                c = _materials.material_constants(attrs, self._eval)
                ms.append((c["mu"], c["lambda"], self._eval(act.attrib.get("phi" + post, "0"))))
            (mu1, l1, p1), (mu2, l2, p2) = ms
            k1, k2 = l1 + 2.0 / 3.0 * mu1, l2 + 2.0 / 3.0 * mu2
            kl = k2 + p1 * (k1 - k2) * (k2 + 4.0 / 3.0 * mu2) / (k2 + 4.0 / 3.0 * mu2 + p2 * (k1 - k2))
            ku = k1 + p2 * (k2 - k1) * (k1 + 4.0 / 3.0 * mu1) / (k1 + 4.0 / 3.0 * mu1 + p1 * (k2 - k1))
            if ku < kl:
                kl, ku = ku, kl
            mul = mu2 + p1 * (mu1 - mu2) / (1 + 2 * p2 * (mu1 - mu2) / (5 * mu2) + 4 * p2 * (mu1 - mu2) / (15 * k2 + 20 * mu2))
            muu = mu1 + p2 * (mu2 - mu1) / (1 + 2 * p1 * (mu2 - mu1) / (5 * mu1) + 4 * p1 * (mu2 - mu1) / (15 * k1 + 20 * mu1))
            if muu < mul:
                mul, muu = muu, mul
            self._hs_bounds = {"lower": {"K": kl, "mu": mul, "lambda": kl - 2.0 / 3.0 * mul},
                               "upper": {"K": ku, "mu": muu, "lambda": ku - 2.0 / 3.0 * muu}}
            log.info("HS lower bounds: K=%g mu=%g lambda=%g", kl, mul, kl - 2.0 / 3.0 * mul)
            log.info("HS upper bounds: K=%g mu=%g lambda=%g", ku, muu, ku - 2.0 / 3.0 * muu)
            return None
        if name == "write_voxel_data":
            # LSSolver::writeData  F:17076-17126: one tab-separated row per voxel (x slowest): indices, the normal if the
            # project carries normals, one column per material with its volume fraction
            fn = self._attr(act, "filename", None, str)
            if not fn:
                raise RuntimeError("write_voxel_data: filename missing")
            self.init_lss()
            self.init_phase()
            nx, ny, nz = self._lss.shape
            cols = [np.repeat(np.arange(nx), ny * nz), np.tile(np.repeat(np.arange(ny), nz), nx), np.tile(np.arange(nz), nx * ny)]
            head = ["i_x", "i_y", "i_z"]
            if self._normals is not None:
                head += ["n_x", "n_y", "n_z"]
                cols += [np.asarray(self._normals[c]).reshape(-1) for c in range(3)]
            head += list(self._phase_names)
            cols += [np.asarray(self._phi[m]).reshape(-1) for m in range(len(self._phase_names))]
            # (the reference adds a_x a_y a_z when the solver holds an orientation field, F:17091-17093: the projects of this
            # path never do -- the field belongs to the fibre-orientation materials outside SURVEY 8)
            table = np.column_stack(cols)
            with open(fn, "w") as f:
                f.write("\t".join(head))
                # rows in slabs of 64 K voxels: one formatting call each instead of a Python loop per voxel (16.7 M at 256^3)
                fmt = "\t".join(["%d"] * 3 + ["%g"] * (table.shape[1] - 3))
                for r0 in range(0, table.shape[0], 65536):
                    block = table[r0:r0 + 65536]
                    f.write("\n" + "\n".join(fmt % tuple(row) for row in block.tolist()))
            return None
        if name in ("init_fibers", "tune_num_threads"):
            # init_fibers F:25615-25618: placed fibres need no generation (the random generator is out of scope);
            # tune_num_threads F:25774-25780 tunes the OpenMP team of the CPU solver: nothing to tune on the GPU path
            if name == "tune_num_threads":
                self.init_lss()
            log.info("action <%s> has nothing to do on the MI355X path: skipped", name)
            return None
        if name == "python":
            self._exec_python(act.text or "")
            return None
        if name == "print_A2":   # F:25747-25752
            log.info("A2: %s", self.get_A2())
            return None
        if name == "print_timings":
            log.info("solve time: %g s", self._lss.solve_time if self._lss else 0.0)
            return None
        if name == "exit":
            return "exit"
        raise RuntimeError("Unknown action: '%s'" % name)

    def _read_raw_data(self, act):
        """read_raw_data  F:25494-25573 + readRawPhase F:16925-17001 + initMultiphase F:16761-16922"""
        self.init_lss()
        shape = self._lss.shape
        n = self._attr(act, "n", 0, int)
        dims = [self._attr(act, k, n if n > 0 else shape[i], int) for i, k in enumerate(("nx", "ny", "nz"))]
        filename = self._attr(act, "filename", None, str)
        if filename is None:
            raise RuntimeError("Undefined property: filename!")
        dtype = self._attr(act, "dtype", "uint8", str)
        treshold = self._attr(act, "treshold", -1.0)
        col_order = self._attr(act, "order", "col", str) == "col"
        header = self._attr(act, "header_bytes", 0, int)
        np_types = {"uint8": (np.uint8, 1 / 255.0), "uint16": (np.uint16, 1 / 65535.0),
                    "uint32": (np.uint32, 1 / 4294967295.0), "float": (np.float32, 1.0), "double": (np.float64, 1.0)}
        if dtype not in np_types:
            raise RuntimeError("Unknown data type '%s'" % dtype)
        nt, default_scale = np_types[dtype]
        scale = self._attr(act, "scale", default_scale)
        opener = gzip.open if filename.endswith(".gz") else open
        try:
            with opener(filename, "rb") as f:
                f.read(header)
                count = dims[0] * dims[1] * dims[2]
                raw = np.frombuffer(f.read(count * np.dtype(nt).itemsize), dtype=nt)
        except OSError as e:
            raise RuntimeError("Error reading file '%s': %s" % (filename, e))
        if raw.size != dims[0] * dims[1] * dims[2]:
            raise RuntimeError("Error reading raw data: file too short")
        if col_order:   # x fastest in the file  F:16946-16966
            t = raw.reshape(dims[2], dims[1], dims[0]).transpose(2, 1, 0)
        else:
            t = raw.reshape(dims[0], dims[1], dims[2])
        t = np.minimum(np.maximum(scale * t.astype(np.float64), 0.0), 1.0)
        if treshold >= 0:
            t = (t > treshold).astype(np.float64)
        if not self._raw_phase:
            self._phi = np.ones((len(self._phase_names),) + shape)  # setPhasesOne
            self._raw_phase = True
        mapping = {}
        single = None
        for k, v in act.attrib.items():
            if k.startswith("material_"):
                mapping[int(k[len("material_"):])] = self._material_id(v)
            elif k == "material":
                single = self._material_id(v)
        if any(dims[i] % shape[i] for i in range(3)):
            raise RuntimeError("Phase dimensions are incompatible %d %d %d %d %d %d"
                               % (dims[0], shape[0], dims[1], shape[1], dims[2], shape[2]))
        s = [max(1, dims[i] // shape[i]) for i in range(3)]
        blocks = t.reshape(shape[0], s[0], shape[1], s[1], shape[2], s[2])
        if single is not None:
            self._phi[single] = blocks.mean(axis=(1, 3, 5))   # psum/ns  F:16838-16843
        elif mapping:
            cls = np.floor(blocks / scale + 0.5).astype(np.int64)
            known = np.isin(cls, list(mapping))
            if not known.all():
                raise RuntimeError("Material value %d not mapped!" % int(cls[~known].flat[0]))
            ns = s[0] * s[1] * s[2]
            nph = len(self._phase_names)
            counts = np.zeros((nph,) + shape)
            cent = np.zeros((nph, 3) + shape)
            off = [np.arange(s[a]) + 0.5 for a in range(3)]
            for m in range(nph):
                vals = [v for v, mm in mapping.items() if mm == m]
                if not vals:
                    continue
                sel = np.isin(cls, vals)
                counts[m] = sel.sum(axis=(1, 3, 5))
                # centroid sums of the fine voxels of class m inside each coarse voxel  F:16826-16832
                cent[m, 0] = (sel * off[0][None, :, None, None, None, None]).sum(axis=(1, 3, 5))
                cent[m, 1] = (sel * off[1][None, None, None, :, None, None]).sum(axis=(1, 3, 5))
                cent[m, 2] = (sel * off[2][None, None, None, None, None, :]).sum(axis=(1, 3, 5))
                self._phi[m] = counts[m] / float(ns)
            for m in range(nph):
                if not any(mm == m for mm in mapping.values()):
                    self._phi[m] = 0.0
            # interface normals of the down-sampled data  F:16861-16907: from the voxel centre to the centroid of the
            # material with the largest share; the search starts at the class of the last fine voxel visited
            last_cls = cls[:, -1, :, -1, :, -1]
            cur = np.vectorize(mapping.get)(last_cls).astype(np.int64)
            for m in range(nph):
                take = counts[m] > np.take_along_axis(counts, cur[None], axis=0)[0]
                cur = np.where(take, m, cur)
            cnt = np.take_along_axis(counts, cur[None], axis=0)[0]
            nrm = np.stack([np.take_along_axis(cent[:, a], cur[None], axis=0)[0] / cnt - 0.5 * s[a] for a in range(3)])
            mag = np.sqrt((nrm * nrm).sum(axis=0))
            flat = mag < 1e-9
            if flat.any():
                # "random" normal e_(kk % 3), kk = index in the reference's padded layout (nzp = 2 (nz/2 + 1))
                nzp_ref = 2 * (shape[2] // 2 + 1)
                ii, jj, kk3 = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
                k3 = (ii * shape[1] * nzp_ref + jj * nzp_ref + kk3) % 3
                for a in range(3):
                    nrm[a] = np.where(flat, (k3 == a).astype(np.float64), nrm[a])
                mag = np.where(flat, 1.0, mag)
            self._raw_normals = nrm / mag
        self._phase_valid = False
        return None


def _normalize_phi(phi):
    """normalizePhi  F:17588-17646: walking materials from last to first, each takes
    min(remaining, phi); the volume fractions of a voxel then sum to at most 1."""
    out = np.empty_like(phi)
    rem = np.ones(phi.shape[1:])
    for m in range(phi.shape[0] - 1, -1, -1):
        vol = np.minimum(rem, phi[m])
        out[m] = vol
        rem = rem - vol
    return out
