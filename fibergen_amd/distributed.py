"""Slab-decomposed Lippmann-Schwinger solver: host-side set-up of the driver that lives below the C ABI.

The reference is single-process (OpenMP + threaded FFTW); this is the multi-GPU counterpart of one
LSSolver.  Rank r owns the x-planes [r*nx/P, (r+1)*nx/P) of every field.  The whole loop -- sweeps,
transforms, the two all-to-alls per component, the halo planes, the all-reduced norms and the stop rule -- runs in
libfibergen_amd.so (fibergen_amd/csrc/fg_slab.hip); Python only creates the members and hands them a transport:

    DistributedLSSolver   one member per process (torch.distributed world): RCCL when the process group is `nccl`
                          (the library opens its own RCCL communicator from an id broadcast over the group), a
                          host-staged callback transport when it is `gloo` (tests: several ranks share one GPU).
    SlabGroup             all P members in THIS process on one GPU and one stream (single-GPU tests,
                          bench.py --force-slab): the very same steps, exchanges become device copies.

Arrays given to / returned by a member have the LOCAL shape [ncomp][nx/P][ny][nz]; SlabGroup takes and returns
global arrays and slices them.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .solver import LSSolver

PLAN_A2A_FORWARD, PLAN_A2A_BACKWARD, PLAN_HALO_U, PLAN_HALO_MODULI, PLAN_HALO_TAU = range(5)
BUFFER_NAMES = ["spectrum_x", "spectrum_y", "u", "moduli", "halo_send_lo", "halo_send_hi", "halo_recv_lo", "halo_recv_hi"]


def slab_plan(nx, ny, nz, nranks, rank, what, comp=0):
    """The exchange plan of `rank` (fg_slab_plan; no GPU needed): (ops, self_copy) with ops = list of dicts
    {send, peer, buffer, offset, count} in doubles, self_copy = (src, dst) or None."""
    lib = _lib.load()
    cap = 4 * nranks + 16
    ops = (_lib.FgPlanOp * cap)()
    selfc = (_lib.FgPlanOp * 2)()
    n = lib.fg_slab_plan(int(nx), int(ny), int(nz), int(nranks), int(rank), int(what), int(comp), ops, cap, selfc)
    if n < 0:
        raise RuntimeError("fg_slab_plan: invalid arguments")
    as_dict = lambda o: dict(send=int(o.send), peer=int(o.peer), buffer=int(o.buffer), offset=int(o.offset), count=int(o.count))
    sc = (as_dict(selfc[0]), as_dict(selfc[1])) if selfc[0].count else None
    return [as_dict(ops[i]) for i in range(n)], sc


class SlabMember(LSSolver):
    """One x-slab (fg_create_slab or a handle of fg_slab_group_create); local shapes."""

    def __init__(self, nx, ny, nz, dx=1.0, dy=1.0, dz=1.0, rank=0, nranks=1, device=0, _handle=None):
        self._lib = _lib.load()
        self.nx_global, self.rank, self.nranks = int(nx), int(rank), int(nranks)
        if nx % nranks or ny % nranks:
            raise RuntimeError("slab decomposition needs nx and ny divisible by the number of ranks")
        self.nx, self.ny, self.nz = int(nx) // int(nranks), int(ny), int(nz)
        self.dx, self.dy, self.dz = float(dx), float(dy), float(dz)
        self.device = int(device)
        if _handle is None:
            _handle = self._lib.fg_create_slab(int(nx), int(ny), int(nz), self.dx, self.dy, self.dz, int(device), int(rank),
                                               int(nranks))
            if not _handle:
                raise RuntimeError(self._lib.fg_last_error(None).decode())
        self._h = _handle
        self._cb_keepalive = None
        self._transport_keepalive = None
        self.nphases = 0
        self.scalar = False

    @property
    def transport(self):
        return self._lib.fg_slab_transport(self._h).decode()

    def slab(self, array):
        """This rank's x-slab of a global array [..., nx, ny, nz]."""
        a = np.asarray(array)
        return np.ascontiguousarray(a[..., self.rank * self.nx:(self.rank + 1) * self.nx, :, :])

    # -- transports ---------------------------------------------------------------------------------------------
    def connect_rccl(self, unique_id: bytes):
        if len(unique_id) != 128:
            raise ValueError("the RCCL unique id has 128 bytes")
        self._check(self._lib.fg_slab_connect_rccl(self._h, unique_id))

    def connect_callback(self, exchange, allreduce):
        """exchange(ops): ops = list of (send, peer, ptr, nbytes); allreduce(values: np.ndarray, min_op) in place."""
        def x_tramp(_user, ops, n):
            try:
                exchange([(int(ops[i].send), int(ops[i].peer), int(ops[i].ptr or 0), int(ops[i].bytes)) for i in range(n)])
                return 0
            except Exception:   # noqa: BLE001 -- no exception may cross the C boundary
                import traceback
                traceback.print_exc()
                return 1

        def r_tramp(_user, values, n, min_op):
            try:
                allreduce(np.ctypeslib.as_array(values, shape=(n,)), bool(min_op))
                return 0
            except Exception:   # noqa: BLE001
                import traceback
                traceback.print_exc()
                return 1

        self._transport_keepalive = (_lib.EXCHANGE_FN(x_tramp), _lib.ALLREDUCE_FN(r_tramp))
        self._check(self._lib.fg_slab_connect_callback(self._h, self._transport_keepalive[0], self._transport_keepalive[1], None))


def rccl_unique_id():
    lib = _lib.load()
    buf = ctypes.create_string_buffer(128)
    if lib.fg_comm_unique_id(buf) != 0:
        raise RuntimeError(lib.fg_last_error(None).decode())
    return buf.raw


class _DeviceBuffer:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v3)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 8,), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


class DistributedLSSolver(SlabMember):
    """One member per process of a torch.distributed world (None = default group).

    backend nccl  -> RCCL inside the library (device = LOCAL_RANK unless given; torch's current device is set too);
    backend gloo  -> callback transport: exchanged bytes are staged through the host (ranks may share a GPU).
    Without an initialised process group this is a lone slab (P = 1)."""

    def __init__(self, nx, ny, nz, dx=1.0, dy=1.0, dz=1.0, group=None, device=None, transport=None):
        import os
        dist = None
        try:
            import torch.distributed as _dist
            if _dist.is_available() and _dist.is_initialized():
                dist = _dist
        except ImportError:
            pass
        rank = dist.get_rank(group) if dist else 0
        nranks = dist.get_world_size(group) if dist else 1
        backend = dist.get_backend(group) if dist else None
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
        super().__init__(nx, ny, nz, dx, dy, dz, rank=rank, nranks=nranks, device=device)
        self._dist, self.group = dist, group
        if nranks == 1:
            return
        # transport = "rccl" (or FG_SLAB_TRANSPORT=rccl) uses the library's RCCL transport whatever torch's backend is: the
        # unique id then travels over gloo (tests: several RCCL ranks on ONE GPU, see tests/test_gpu_distributed.py)
        transport = transport or os.environ.get("FG_SLAB_TRANSPORT")
        if backend == "nccl" or transport == "rccl":
            import torch
            if backend == "nccl":
                torch.cuda.set_device(self.device)   # torch-side collectives of this process use the same GPU
            box = [rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            self.connect_rccl(box[0])
        else:
            self.connect_callback(self._gloo_exchange, self._gloo_allreduce)

    # host-staged transport (tests): device pointers -> torch views -> CPU tensors over gloo
    def _gloo_exchange(self, ops):
        import torch
        dist = self._dist
        dev = "cuda:%d" % self.device
        reqs, staged = [], []
        for send, peer, ptr, nbytes in ops:
            t = torch.as_tensor(_DeviceBuffer(ptr, nbytes), device=dev)
            g_peer = dist.get_global_rank(self.group, peer) if self.group is not None else peer
            if send:
                reqs.append(dist.P2POp(dist.isend, t.cpu(), g_peer, self.group))
            else:
                c = torch.empty(nbytes // 8, dtype=torch.float64)
                staged.append((t, c))
                reqs.append(dist.P2POp(dist.irecv, c, g_peer, self.group))
        for w in dist.batch_isend_irecv(reqs):
            w.wait()
        for t, c in staged:
            t.copy_(c)
        torch.cuda.synchronize(self.device)

    def _gloo_allreduce(self, values, min_op):
        import torch
        dist = self._dist
        t = torch.from_numpy(values.copy())
        parts = [torch.empty_like(t) for _ in range(self.nranks)]
        dist.all_gather(parts, t, group=self.group)
        acc = parts[0].clone()
        for r in range(1, self.nranks):   # fixed rank order => identical on every rank
            acc = torch.minimum(acc, parts[r]) if min_op else acc + parts[r]
        values[:] = acc.numpy()


class GlobalViewSolver(DistributedLSSolver):
    """A DistributedLSSolver that looks like ONE LSSolver to the project layer (`FG.decompose_slabs`): global arrays in
    (every rank passes the whole field, its x-slab is cut out here), global arrays out (`get_field` gathers the slabs on
    every rank -- meant for results of moderate size), scalars are collective anyway."""

    @property
    def shape(self):
        return (self.nx_global, self.ny, self.nz)

    @property
    def local_shape(self):
        return (self.nx, self.ny, self.nz)

    def set_phase(self, p, mu, lam, phi=None):
        self._check(self._lib.fg_set_phase(self._h, int(p), float(mu), float(lam),
                                           None if phi is None else self.slab(np.asarray(phi, dtype=np.float64)).ctypes.data_as(_lib.c_double_p)))

    def set_normals(self, normals):
        n = self.slab(np.asarray(normals, dtype=np.float64))
        self._check(self._lib.fg_set_normals(self._h, n.ctypes.data_as(_lib.c_double_p)))

    # Largest global field get_field will assemble on every rank (bytes); FG_GATHER_LIMIT_GB overrides.  Beyond it the
    # caller should read the slabs (DistributedLSSolver.get_field with local shapes) instead of P copies of everything.
    GATHER_LIMIT_GB = 16.0      # default of FG_GATHER_LIMIT_GB (read at every call)
    GATHER_CHUNK = 256 * 2 ** 20   # bytes of a rank's slab per all_gather: the staging buffers stay at (P + 1) x this

    def _gather(self, local):
        """The slabs of all ranks, concatenated along x, on every rank: all_gathers of float64 pieces of at most GATHER_CHUNK
        bytes per rank (on this solver's device for an nccl group -- (P + 1) x 256 MB of staging whatever the field's size,
        so a card the solver has filled still serves get_field --, on the host for gloo); no pickling.  The piece count
        follows from the slab's size alone, so every rank issues the same sequence."""
        if self.nranks == 1:
            return local
        import os
        limit = int(float(os.environ.get("FG_GATHER_LIMIT_GB", self.GATHER_LIMIT_GB)) * 2 ** 30)
        total = local.nbytes * self.nranks
        if total > limit:
            raise RuntimeError("get_field: the global field is %.1f GB (limit %.1f GB, FG_GATHER_LIMIT_GB): read the slabs "
                               "rank by rank instead of gathering it on every rank" % (total / 2 ** 30, limit / 2 ** 30))
        import torch
        dist = self._dist
        flat = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64)).reshape(-1)
        on_device = dist.get_backend(self.group) == "nccl"
        dev = torch.device("cuda", int(self.device)) if on_device else torch.device("cpu")
        n = flat.numel()
        step = max(1, self.GATHER_CHUNK // 8)
        parts = np.empty((self.nranks, n))
        for lo in range(0, n, step):
            piece = flat[lo:lo + step].to(dev)
            out = torch.empty(self.nranks * piece.numel(), dtype=piece.dtype, device=dev)   # flat: the form gloo and nccl both take
            dist.all_gather_into_tensor(out, piece, group=self.group)
            parts[:, lo:lo + piece.numel()] = out.cpu().numpy().reshape(self.nranks, -1)
        return np.concatenate(list(parts.reshape((self.nranks,) + tuple(local.shape))), axis=1)

    def get_field(self, name):
        nc = self._lib.fg_field_components(self._h, name.encode())
        out = np.empty((nc,) + self.local_shape)
        self._check(self._lib.fg_get_field(self._h, name.encode(), out.ctypes.data_as(_lib.c_double_p)))
        return self._gather(out)

    def set_field(self, name, value):
        v = self.slab(np.ascontiguousarray(value, dtype=np.float64))
        self._check(self._lib.fg_set_field(self._h, name.encode(), v.ctypes.data_as(_lib.c_double_p)))


class SlabGroup:
    """All P slabs of one problem in this process on ONE GPU (fg_slab_group_create).  Global arrays in and out;
    every collective call drives all members through the same steps the multi-GPU run executes."""

    def __init__(self, nx, ny, nz, dx=1.0, dy=1.0, dz=1.0, nranks=1, device=0):
        lib = _lib.load()
        handles = (ctypes.c_void_p * nranks)()
        if lib.fg_slab_group_create(int(nx), int(ny), int(nz), float(dx), float(dy), float(dz), int(device), int(nranks),
                                    handles) != 0:
            raise RuntimeError(lib.fg_last_error(None).decode())
        self.members = [SlabMember(nx, ny, nz, dx, dy, dz, rank=r, nranks=nranks, device=device, _handle=handles[r])
                        for r in range(nranks)]
        self.nx, self.ny, self.nz, self.nranks = int(nx), int(ny), int(nz), int(nranks)

    def close(self):
        for m in self.members:
            m.close()
        self.members = []

    # configuration: the same on every member, fields sliced
    def set_num_phases(self, n):
        for m in self.members:
            m.set_num_phases(n)

    def set_phase(self, p, mu, lam, phi=None):
        for m in self.members:
            m.set_phase(p, mu, lam, None if phi is None else m.slab(phi))

    def set_normals(self, normals):
        for m in self.members:
            m.set_normals(m.slab(normals))

    def set_options(self, **kw):
        for m in self.members:
            m.set_options(**kw)

    def set_bc_projector(self, P):
        for m in self.members:
            m.set_bc_projector(P)

    def set_convergence_callback(self, fn):
        self.members[0].set_convergence_callback(fn)

    def set_field(self, name, value):
        for m in self.members:
            m.set_field(name, m.slab(value))

    # collective calls: any member drives the group
    def run(self, E, S=None):
        return self.members[0].run(E, S)

    def run_load_steps(self, E, S=None, params=(0.0, 1.0), first=None, step_callback=None):
        return self.members[0].run_load_steps(E, S, params, first, step_callback)

    def iterate(self, E, n):
        self.members[0].iterate(E, n)

    def time_iterations(self, E, n):
        return self.members[0].time_iterations(E, n)

    def mean_stress(self):
        return self.members[0].mean_stress()

    def mean_strain(self):
        return self.members[0].mean_strain()

    def volume_fraction(self, p):
        return self.members[0].volume_fraction(p)

    def calc_ref_material(self):
        return self.members[0].calc_ref_material()

    def synchronize(self):
        self.members[0].synchronize()

    def get_field(self, name):
        return np.concatenate([m.get_field(name) for m in self.members], axis=1)

    @property
    def iterations(self):
        return self.members[0].iterations

    @property
    def residuals(self):
        return self.members[0].residuals

    @property
    def solve_time(self):
        return self.members[0].solve_time

    @property
    def ref_material(self):
        return self.members[0].ref_material
