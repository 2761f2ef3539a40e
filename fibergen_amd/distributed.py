"""Slab-decomposed Lippmann-Schwinger solver: one process per GPU, x-slabs, RCCL over xGMI.

The reference is single-process (OpenMP + threaded FFTW); this is the multi-GPU
counterpart of one LSSolver.  Rank r owns the x-planes [r*nx/P, (r+1)*nx/P) of
every field.  Per basicScheme pass (F:20558-20578):

    phase 0  polarisation (local)                                  | halo: tau0 -> right, tau5,tau4 -> left
    phase 1  divergence, z-r2c, y-c2c, pack                        | all-to-all  (x-slabs -> y-slabs)
    phase 2  x-c2c, Green operator on the y-slab, x-c2c^-1, pack   | all-to-all  (y-slabs -> x-slabs)
    phase 3  y-c2c^-1, z-c2r                                       | halo: u1,u2 -> right, u0 -> left
    phase 4  strain operator + local sums of squares               | all-gather of 6 doubles

Two all-to-alls per iteration with the three components batched into one message per peer
(each GPU sends a distinct 1/P block to every peer over its own xGMI link), two +-1-plane
halo exchanges, and tiny all-gathers for the norms / means, summed in rank order so every
rank takes identical stop decisions.

The compute backend is `HipSlabBackend` (the C ABI's fg_slab_phase on this rank's GPU).
The driver only needs `phase / buffer / local_sums`, so the CPU test-suite can drive it
with a NumPy stand-in over gloo (tests/test_distributed_cpu.py).
"""
from __future__ import annotations

import ctypes
import math
import time

import numpy as np

EPS = np.finfo(np.float64).eps
SMALLEST = np.finfo(np.float64).tiny


# --------------------------------------------------------------------------- Voigt helpers (host)
def _voigt_id4():
    return np.diag([1.0, 1.0, 1.0, 0.5, 0.5, 0.5])


def _voigt_mv(M, v):
    vc = np.array(v, dtype=np.float64)
    vc[3:6] *= 2
    return M @ vc


def _voigt_mm(A, B):
    return np.stack([_voigt_mv(A, B[:, i]) for i in range(6)], axis=1)


def _voigt_norm2(v):
    v = np.asarray(v, dtype=np.float64)
    return math.sqrt(float(v @ v) + v[3] ** 2 + v[4] ** 2 + v[5] ** 2)


def bc_matrices(P, mu_0, lambda_0):
    """setBCProjector  F:20599-20665 -> (Q, QC0, M, MQ)."""
    Q = _voigt_id4() - P
    if not np.any(Q):
        Z = np.zeros((6, 6))
        return Q, Z, Z.copy(), Z.copy()
    II = np.zeros((6, 6))
    II[:3, :3] = 1.0
    C0 = 2 * mu_0 * _voigt_id4() + lambda_0 * II
    QC0 = _voigt_mm(Q, C0)
    if math.isnan(mu_0):
        N = np.full((6, 6), np.nan)
        return Q, QC0, N, N.copy()
    QC0Q = _voigt_mm(QC0, Q)
    A = np.empty((9, 9))
    for i in range(9):
        for j in range(i, 9):
            A[j, i] = A[i, j] = QC0Q[i if i < 6 else i - 3, j if j < 6 else j - 3]
    w, V = np.linalg.eigh(A)
    thr = math.sqrt(EPS) * np.linalg.norm(w)
    winv = np.array([1.0 / x if abs(x) > thr else 0.0 for x in w])
    M = (V * winv) @ V.T
    for i in range(3):
        for j in range(6):
            M[j, 3 + i] = 0.5 * (M[j, 3 + i] + M[j, 6 + i])
        for j in range(6):
            M[3 + i, j] = 0.5 * (M[3 + i, j] + M[6 + i, j])
    M = M[:6, :6].copy()
    return Q, QC0, M, _voigt_mm(M, Q)


# --------------------------------------------------------------------------- HIP backend
class _DeviceBuffer:
    """Zero-copy view of a device allocation for torch.as_tensor (CUDA array interface v3)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


class HipSlabBackend:
    """This rank's x-slab on its GPU, through the C ABI (fg_create_slab / fg_slab_phase)."""

    def __init__(self, nx, ny, nz, dx, dy, dz, rank, nranks, device):
        from . import _lib
        from .solver import LSSolver
        self._lib = _lib.load()
        s = LSSolver.__new__(LSSolver)
        s._lib = self._lib
        s.nx, s.ny, s.nz = nx // nranks, ny, nz          # local shape for field I/O
        s.dx, s.dy, s.dz = float(dx), float(dy), float(dz)
        s._h = self._lib.fg_create_slab(int(nx), int(ny), int(nz), float(dx), float(dy), float(dz), int(device),
                                        int(rank), int(nranks))
        if not s._h:
            raise RuntimeError(self._lib.fg_last_error(None).decode())
        s._cb_keepalive = None
        s.nphases = 0
        self.solver = s
        self.device = device
        self._buffers = {}

    def phase(self, k, E=None, R=None):
        dp = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(
            ctypes.POINTER(ctypes.c_double))
        Ea = None if E is None else np.ascontiguousarray(E, dtype=np.float64)
        Ra = None if R is None else np.ascontiguousarray(R, dtype=np.float64)
        self.solver._check(self._lib.fg_slab_phase(self.solver._h, int(k), dp(Ea), dp(Ra)))

    def synchronize(self):
        self.solver.synchronize()

    def buffer(self, name):
        """torch tensor (float64, 1-D) aliasing the named exchange buffer on the GPU."""
        if name not in self._buffers:
            import torch
            nbytes = ctypes.c_ulong(0)
            ptr = self._lib.fg_exchange_buffer(self.solver._h, name.encode(), ctypes.byref(nbytes))
            if not ptr:
                raise RuntimeError(self._lib.fg_last_error(self.solver._h).decode())
            self._buffers[name] = torch.as_tensor(_DeviceBuffer(ptr, nbytes.value), device="cuda:%d" % self.device)
        return self._buffers[name]

    def local_sums(self, what):
        n = {"tangent_minmax": 2}.get(what, 1 if what.startswith("phi:") else 6)
        out = np.zeros(n)
        self.solver._check(self._lib.fg_local_sums(self.solver._h, what.encode(),
                                                   out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return out

    # configuration / field I/O: forwarded to the local solver object (local shapes)
    def __getattr__(self, name):
        return getattr(self.solver, name)


# --------------------------------------------------------------------------- the driver
class DistributedLSSolver:
    """LSSolver surface (materials, phases, options, run, means, fields) over P slabs.

    Arrays given to / returned by set_phase, set_normals, get_field, set_field are LOCAL slabs
    [ncomp][nx/P][ny][nz]; `slab()` cuts a global array.  `group` is a torch.distributed
    process group (None = default); with P == 1 no communication library is touched.
    """

    def __init__(self, nx, ny, nz, dx=1.0, dy=1.0, dz=1.0, backend=None, group=None, rank=None, nranks=None,
                 device=None):
        self._dist = None
        if nranks is None:
            try:
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    self._dist = dist
                    nranks = dist.get_world_size(group)
                    rank = dist.get_rank(group)
            except ImportError:
                pass
            if nranks is None:
                nranks, rank = 1, 0
        elif nranks > 1:
            import torch.distributed as dist
            self._dist = dist
        self.group = group
        self.rank, self.nranks = int(rank), int(nranks)
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        if self.nx % self.nranks or self.ny % self.nranks:
            raise RuntimeError("slab decomposition needs nx and ny divisible by the number of ranks")
        self.nxl = self.nx // self.nranks
        self.N = self.nx * self.ny * self.nz
        if backend is None:
            if device is None:
                import os
                device = int(os.environ.get("LOCAL_RANK", "0"))
            backend = HipSlabBackend(nx, ny, nz, dx, dy, dz, self.rank, self.nranks, device)
        self.backend = backend
        # solver state mirrored from SolverOptions (F:14800-14862)
        self.tol, self.abs_tol, self.bc_tol, self.maxiter = 1e-4, EPS, 1e-3, 10000
        self.ref_scale, self.bc_relax, self.update_ref = 1.0, 1.0, True
        self.mu_0, self.lambda_0 = float("nan"), 0.0
        self.BC_P = _voigt_id4()
        self.callback = None
        self.residuals = []
        self.iterations = 0
        self.solve_time = 0.0
        self.comm_time = 0.0

    # -- helpers ---------------------------------------------------------------
    def slab(self, array):
        """This rank's x-slab of a global array [..., nx, ny, nz]."""
        a = np.asarray(array)
        return np.ascontiguousarray(a[..., self.rank * self.nxl:(self.rank + 1) * self.nxl, :, :])

    @property
    def shape(self):
        return (self.nxl, self.ny, self.nz)

    def set_num_phases(self, n):
        self.backend.set_num_phases(n)

    def set_phase(self, p, mu, lam, phi_local=None):
        self.backend.set_phase(p, mu, lam, phi_local)

    def set_normals(self, normals_local):
        self.backend.set_normals(normals_local)

    def set_options(self, **kw):
        for k in ("tol", "abs_tol", "bc_tol", "ref_scale", "bc_relax", "mu_0", "lambda_0"):
            if k in kw:
                setattr(self, k, float(kw[k]))
        if "maxiter" in kw:
            self.maxiter = int(kw["maxiter"])
        if "update_ref" in kw:
            self.update_ref = kw["update_ref"] not in ("never", 0, False)
        fwd = {k: v for k, v in kw.items() if k in ("mixing_rule", "eps_g", "eps_a", "mu_0", "lambda_0", "fuse_x")}
        if fwd:
            self.backend.set_options(**fwd)

    def set_bc_projector(self, P):
        P = np.asarray(P, dtype=np.float64)
        se = math.sqrt(EPS)
        if P.shape != (6, 6) or np.linalg.norm(P - P.T) > se:
            raise RuntimeError("Projector is not symmetric")
        if np.linalg.norm(P - _voigt_mm(P, P)) > se:
            raise RuntimeError("Specified Projector is not a projector")
        self.BC_P = P

    def set_convergence_callback(self, fn):
        self.callback = fn

    def get_field(self, name):
        return self.backend.get_field(name)

    def set_field(self, name, value):
        self.backend.set_field(name, value)

    # -- communication ---------------------------------------------------------------
    def _comm_tensor(self, t):
        """Tensor the process group can move: device tensors as they are under NCCL/RCCL,
        staged through the host under gloo."""
        if t.is_cuda and self._dist.get_backend(self.group) != "nccl":
            return t.cpu()
        return t

    def _exchange(self, sends, recvs):
        """sends / recvs: lists of (tensor, peer).  Point-to-point, all posted at once."""
        dist = self._dist
        t0 = time.perf_counter()
        self.backend.synchronize()
        ops, staged = [], []
        for (t, peer) in recvs:
            c = self._comm_tensor(t)
            staged.append((t, c))
            ops.append(dist.P2POp(dist.irecv, c, peer, self.group))
        for (t, peer) in sends:
            ops.append(dist.P2POp(dist.isend, self._comm_tensor(t), peer, self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for t, c in staged:
            if c is not t:
                t.copy_(c)
        if staged and staged[0][0].is_cuda:
            import torch
            torch.cuda.synchronize()
        self.comm_time += time.perf_counter() - t0

    @staticmethod
    def _sync_copies(t):
        # torch copies run on torch's stream, the next phase on the solver's: order them on the host
        if t.is_cuda:
            import torch
            torch.cuda.synchronize()

    def _halo(self):
        b = self.backend
        s_lo, s_hi, r_lo, r_hi = (b.buffer(n) for n in ("halo_send_lo", "halo_send_hi", "halo_recv_lo", "halo_recv_hi"))
        if self.nranks == 1:
            self.backend.synchronize()
            r_lo.copy_(s_hi)   # my last plane is my own x-1 neighbour (periodic)
            r_hi.copy_(s_lo)
            self._sync_copies(r_lo)
            return
        left, right = (self.rank - 1) % self.nranks, (self.rank + 1) % self.nranks
        self._exchange([(s_hi, right), (s_lo, left)], [(r_lo, left), (r_hi, right)])

    def _all_to_all(self):
        b = self.backend
        send, recv = b.buffer("a2a_send"), b.buffer("a2a_recv")
        P = self.nranks
        blk = send.numel() // P
        if P == 1:
            self.backend.synchronize()
            recv.copy_(send)
            self._sync_copies(recv)
            return
        me = self.rank
        self.backend.synchronize()
        recv[me * blk:(me + 1) * blk].copy_(send[me * blk:(me + 1) * blk])
        sends = [(send[q * blk:(q + 1) * blk], q) for q in range(P) if q != me]
        recvs = [(recv[q * blk:(q + 1) * blk], q) for q in range(P) if q != me]
        self._exchange(sends, recvs)

    def _gather(self, vec):
        """[P, len(vec)] array of every rank's vector (rank order)."""
        vec = np.asarray(vec, dtype=np.float64)
        if self.nranks == 1:
            return vec[None, :]
        import torch
        dist = self._dist
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor(vec, dtype=torch.float64, device=dev)
        out = [torch.empty_like(t) for _ in range(self.nranks)]
        dist.all_gather(out, t, group=self.group)
        return np.stack([o.cpu().numpy() for o in out])

    def _global_sum(self, what):
        parts = self._gather(self.backend.local_sums(what))
        total = parts[0].copy()
        for r in range(1, self.nranks):   # fixed rank order => identical on every rank
            total = total + parts[r]
        return total

    # -- reductions ---------------------------------------------------------------------
    def mean_stress(self):
        return self._global_sum("stress")

    def mean_strain(self):
        return self._global_sum("epsilon") / float(self.N)

    def volume_fraction(self, p):
        return float(self._global_sum("phi:%d" % p)[0]) / float(self.N)

    def calc_ref_material(self):
        """calcRefMaterial  F:22283-22313 with the min / max taken over all slabs."""
        parts = self._gather(self.backend.local_sums("tangent_minmax"))
        lo, hi = float(parts[:, 0].min()), float(parts[:, 1].max())
        if lo < 0:
            lo = 0.0
        mu_0 = 0.5 * (lo + hi)
        mu_0 *= 0.5 * self.ref_scale
        self.mu_0 = mu_0
        self.backend.set_options(mu_0=self.mu_0, lambda_0=self.lambda_0)
        return self.mu_0, self.lambda_0

    # -- one pass ---------------------------------------------------------------------------
    def basic_scheme(self, E, bc=None):
        """One basicScheme pass over all slabs; returns the six global sums of squares."""
        b = self.backend
        Q, QC0, M, MQ = bc if bc is not None else bc_matrices(self.BC_P, self.mu_0, self.lambda_0)
        F00 = self.mean_strain() if self.bc_relax != 1.0 else np.zeros(6)
        b.phase(0)
        mq_zero = np.linalg.norm(MQ) < EPS
        F0 = np.zeros(6) if mq_zero else self._global_sum("tau") / float(self.N)
        self._halo()
        b.phase(1)
        self._all_to_all()
        b.phase(2)
        self._all_to_all()
        b.phase(3)
        self._halo()
        R = None
        if not (mq_zero and self.bc_relax == 1.0):
            R = -1.0 * (self.bc_relax * _voigt_mv(MQ, F0) - (1 - self.bc_relax) * _voigt_mv(M, _voigt_mv(QC0, F00)))
        b.phase(4, E, R)
        return self._global_sum("sumsq")

    def iterate(self, E, n):
        bc = bc_matrices(self.BC_P, self.mu_0, self.lambda_0)
        for _ in range(n):
            self.basic_scheme(E, bc)

    # -- LSSolver::run  F:21247-21398 / runBasic F:21716-21805 / _converged F:21177-21244 -----
    def run(self, E0, S0=None):
        E0 = np.asarray(E0, dtype=np.float64)
        S0 = np.zeros(6) if S0 is None else np.asarray(S0, dtype=np.float64)
        self.residuals = []
        t_start = time.perf_counter()
        Q, QC0, M, MQ = bc_matrices(self.BC_P, self.mu_0, self.lambda_0)
        se = math.sqrt(EPS)
        if np.linalg.norm(_voigt_mv(self.BC_P, S0)) > se * np.linalg.norm(S0):
            raise RuntimeError("Incompatible stress boundary condition specified")
        if np.linalg.norm(_voigt_mv(Q, E0)) > se * np.linalg.norm(E0):
            raise RuntimeError("Incompatible strain boundary condition specified")
        self.backend.set_field("epsilon", np.zeros((6,) + self.shape))   # F:21379
        prev = 0.0
        it = 1
        update_ref = self.update_ref
        E = E0
        bc = (Q, QC0, M, MQ)
        failed = False
        while True:
            if update_ref:
                self.calc_ref_material()
                bc = bc_matrices(self.BC_P, self.mu_0, self.lambda_0)
                E = E0 + self.bc_relax * _voigt_mv(bc[2], S0 - _voigt_mv(bc[1], E0))   # calcBCMean F:20242
                update_ref = False
            sumsq = self.basic_scheme(E, bc)
            m = np.sqrt(sumsq / float(self.N))
            cur = math.sqrt(float((m * m).sum() + (m[3:] * m[3:]).sum()))
            abs_err = abs(prev - cur)
            rel_err = abs_err / (SMALLEST + cur)
            prev = cur
            if math.isnan(rel_err):
                failed = True
                break
            self.residuals.append(rel_err)
            if self.callback is not None and self.callback():
                break
            if it >= self.maxiter:
                break
            if rel_err <= self.tol or abs_err <= self.abs_tol:
                if self._bc_error(E0, S0, bc[0]) <= self.bc_tol:
                    break
            it += 1
        self.iterations = it
        self.solve_time = time.perf_counter() - t_start
        return failed

    def _bc_error(self, E_cur, S_cur, Q):
        """bc_error  F:21129-21161"""
        Emean, Smean = self.mean_strain(), self.mean_stress()
        PE = _voigt_mv(self.BC_P, Emean)
        QS = _voigt_mv(Q, Smean)
        norm_E = _voigt_norm2(_voigt_mv(self.BC_P, E_cur))
        err_F = _voigt_norm2(PE - E_cur) / (1 if norm_E < self.bc_tol else norm_E)
        norm_S = _voigt_norm2(S_cur)
        err_S = _voigt_norm2(QS - S_cur) / (1 if norm_S < self.bc_tol else norm_S)
        return max(err_F, err_S)
