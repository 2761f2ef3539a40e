"""Worker for the multi-rank tests: run under torch.distributed.run with 2+ ranks over gloo.
    --backend plan : CPU suite.  Executes the library's exchange plan (fg_slab_plan: all-to-all blocks of the pencil
                     transpose, halo planes) on NumPy buffers and checks every received value against the global array
                     it must come from -- the offsets / peers RCCL will be handed on the GPUs.
    --backend hip  : GPU suite on a 1-GPU box.  The slab driver of libfibergen_amd.so, one member per process, ranks
                     share the GPU, exchanged bytes are staged through the host (callback transport).
Each rank writes its results to --out.<rank>.npz."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run_plan(a, dist, rank, P, grid):
    import torch
    from fibergen_amd.distributed import (PLAN_A2A_BACKWARD, PLAN_A2A_FORWARD, PLAN_HALO_MODULI, PLAN_HALO_TAU, PLAN_HALO_U,
                                          slab_plan)
    nx, ny, nz = grid
    nxl, nyl = nx // P, ny // P
    nzf = nz // 2 + 1
    nzc = ((nzf + 7) // 8) * 8 if nz >= 64 else nzf
    nzp = 2 * nzc
    n = nxl * ny * nzp
    plane = ny * nzp
    ucs = n + 4 * plane
    # global tagged array: value encodes (component, x, y, z') uniquely
    c_, x_, y_, z_ = np.meshgrid(np.arange(3), np.arange(nx), np.arange(ny), np.arange(nzp), indexing="ij")
    G = (((c_ * nx + x_) * ny + y_) * nzp + z_).astype(np.float64)

    def execute(ops, selfc, bufs):
        if selfc is not None:
            s, d = selfc
            bufs[d["buffer"]][d["offset"]:d["offset"] + d["count"]] = bufs[s["buffer"]][s["offset"]:s["offset"] + s["count"]]
        reqs, staged = [], []
        for o in ops:
            view = bufs[o["buffer"]][o["offset"]:o["offset"] + o["count"]]
            if o["send"]:
                reqs.append(dist.P2POp(dist.isend, torch.from_numpy(view.copy()), o["peer"]))
            else:
                t = torch.empty(o["count"], dtype=torch.float64)
                staged.append((view, t))
                reqs.append(dist.P2POp(dist.irecv, t, o["peer"]))
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
        for view, t in staged:
            view[:] = t.numpy()

    errors = []
    # ---- pencil transpose: blocked x-slab spectrum -> y-slab and back
    Gs = G[:, rank * nxl:(rank + 1) * nxl]                               # my x-slab [3][nxl][ny][nzp]
    S = Gs.reshape(3, nxl, P, nyl, nzp).transpose(0, 2, 1, 3, 4).copy()  # blocked layout [3][q][nxl][nyl][nzp]
    bufs = {0: S.reshape(-1).copy(), 1: np.full(3 * n, -1.0)}
    for c in range(3):
        ops, selfc = slab_plan(nx, ny, nz, P, rank, PLAN_A2A_FORWARD, c)
        execute(ops, selfc, bufs)
    R = bufs[1].reshape(3, nx, nyl, nzp)                                  # y-slab [3][nx][nyl][nzp]
    if not np.array_equal(R, G[:, :, rank * nyl:(rank + 1) * nyl]):
        errors.append("forward all-to-all")
    bufs[0][:] = -1.0
    for c in range(3):
        ops, selfc = slab_plan(nx, ny, nz, P, rank, PLAN_A2A_BACKWARD, c)
        execute(ops, selfc, bufs)
    if not np.array_equal(bufs[0], S.reshape(-1)):
        errors.append("backward all-to-all")
    # ---- the same with the three components of a peer in ONE message (comp = -1): x-slab side [q][c][nxl][nyl][nzp],
    #      y-slab side [p][c][nxl][nyl][nzp]
    Si = Gs.reshape(3, nxl, P, nyl, nzp).transpose(2, 0, 1, 3, 4).copy()   # [q][c][nxl][nyl][nzp]
    bufs = {0: Si.reshape(-1).copy(), 1: np.full(3 * n, -1.0)}
    ops, selfc = slab_plan(nx, ny, nz, P, rank, PLAN_A2A_FORWARD, -1)
    if len(ops) != 2 * (P - 1):
        errors.append("one message per peer")
    execute(ops, selfc, bufs)
    Ri = bufs[1].reshape(P, 3, nxl, nyl, nzp)                               # block p = planes [p nxl, (p+1) nxl) of my ky rows
    want = G[:, :, rank * nyl:(rank + 1) * nyl].reshape(3, P, nxl, nyl, nzp).transpose(1, 0, 2, 3, 4)
    if not np.array_equal(Ri, want):
        errors.append("forward all-to-all, interleaved")
    bufs[0][:] = -1.0
    ops, selfc = slab_plan(nx, ny, nz, P, rank, PLAN_A2A_BACKWARD, -1)
    execute(ops, selfc, bufs)
    if not np.array_equal(bufs[0], Si.reshape(-1)):
        errors.append("backward all-to-all, interleaved")
    # ---- halo planes of u (3 components) and of the moduli (2)
    for what, nc, buf_id in ((PLAN_HALO_U, 3, 2), (PLAN_HALO_MODULI, 2, 3)):
        U = np.full((nc, nxl + 4, ny, nzp), -1.0)
        U[:, :nxl] = Gs[:nc]
        bufs = {buf_id: U.reshape(-1)}
        ops, selfc = slab_plan(nx, ny, nz, P, rank, what)
        execute(ops, selfc, bufs)
        U = bufs[buf_id].reshape(nc, nxl + 4, ny, nzp)
        right0 = G[:nc, ((rank + 1) % P) * nxl]                           # first plane of the right neighbour
        left_last = G[:nc, ((rank - 1) % P) * nxl + nxl - 1]              # last plane of the left neighbour
        if not (np.array_equal(U[:, nxl], right0) and np.array_equal(U[:, nxl + 3], left_last)):
            errors.append("halo planes %d" % what)
        if not np.array_equal(U[:, :nxl], Gs[:nc]) or (U[:, nxl + 1:nxl + 3] != -1.0).any():
            errors.append("halo exchange touched other planes %d" % what)
    # ---- halo of tau: tau0 of plane -1 | tau5, tau4 of plane nxl (6 components tagged through G's first three twice)
    T = np.concatenate([Gs, Gs + 0.5])                                     # tau0..tau5 of my slab
    Tg = np.concatenate([G, G + 0.5])
    bufs = {4: np.concatenate([T[5, 0].reshape(-1), T[4, 0].reshape(-1)]), 5: np.concatenate([T[0, nxl - 1].reshape(-1), np.zeros(plane)]),
            6: np.full(2 * plane, -1.0), 7: np.full(2 * plane, -1.0)}
    ops, selfc = slab_plan(nx, ny, nz, P, rank, PLAN_HALO_TAU)
    execute(ops, selfc, bufs)
    xl, xr = ((rank - 1) % P) * nxl + nxl - 1, ((rank + 1) % P) * nxl
    if not (np.array_equal(bufs[6][:plane], Tg[0, xl].reshape(-1)) and np.array_equal(bufs[7][:plane], Tg[5, xr].reshape(-1))
            and np.array_equal(bufs[7][plane:], Tg[4, xr].reshape(-1))):
        errors.append("halo of tau")
    np.savez(a.out + ".%d.npz" % rank, errors=np.array(errors))


def fg_slabs_project(grid, tol, method, mixing, mode):
    """the XML project of the fg-slabs backend (also what the test process runs on one GPU)"""
    mats = ('<matrix E="1" nu="0.3" /><inclusion E="10" nu="0.2" />' if mode == "elasticity" else
            '<matrix mu="1" /><inclusion mu="%s" />' % ("0.05" if mode == "viscosity" else "10"))
    return """<settings><solver nx="%d" ny="%d" nz="%d"><mode>%s</mode><tol>%g</tol><method>%s</method><mixing_rule>%s</mixing_rule>
      <materials>%s</materials></solver>
      <actions><select_material name="inclusion" /><place_fiber R="0.3" /><init_phase normals="1" /><calc_effective_properties /></actions>
    </settings>""" % (grid[0], grid[1], grid[2], mode, tol, method, mixing, mats)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="plan")
    ap.add_argument("--grid", default="8,8,6")
    ap.add_argument("--dims", default="1,1,1")
    ap.add_argument("--mixing", default="voigt")
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--mixed-bc", type=int, default=0)
    ap.add_argument("--split", type=int, default=-1)
    ap.add_argument("--method", default="basic")
    ap.add_argument("--mode", default="elasticity")
    ap.add_argument("--out", required=True)
    ap.add_argument("--stop-rank", type=int, default=-1, help="only this rank installs a convergence callback ...")
    ap.add_argument("--stop-iter", type=int, default=0, help="... which asks to stop in this iteration")
    ap.add_argument("--step-stop-rank", type=int, default=-1,
                    help="load stepping over [0, 0.5, 1]: only this rank installs a load-step callback, which asks to stop "
                         "after step 1 (ADVICE r3: the answer must take every rank along)")
    ap.add_argument("--cancel-rank", type=int, default=-1,
                    help="this rank calls fg_cancel from another thread a little while into an endless run (tol 0)")
    ap.add_argument("--bad-rank", type=int, default=-1,
                    help="laminate mixing: this rank's slab gets one voxel with THREE phases (a device-side error that only "
                         "this rank sees locally)")
    ap.add_argument("--transport", default="",
                    help="rccl: the library's RCCL transport between the ranks although they share ONE GPU -- every rank poses "
                         "as a host of its own (NCCL_HOSTID), so that RCCL connects them through its socket transport on the "
                         "loop-back interface instead of refusing the duplicate device")
    a = ap.parse_args()
    if a.transport == "rccl":
        os.environ["NCCL_HOSTID"] = "fibergen-test-rank-%s" % os.environ.get("RANK", "0")
        os.environ["NCCL_SOCKET_IFNAME"] = "lo"
        os.environ["NCCL_IB_DISABLE"] = "1"
        os.environ.setdefault("NCCL_DEBUG", "WARN")
    import torch  # noqa: F401  before the HIP library: one shared runtime
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    grid = tuple(int(v) for v in a.grid.split(","))
    dims = tuple(float(v) for v in a.dims.split(","))
    if a.backend == "plan":
        run_plan(a, dist, rank, P, grid)
        dist.barrier()
        dist.destroy_process_group()
        return
    if a.backend == "gather":
        # GlobalViewSolver._gather (the all-gather behind get_field) on a stand-in object: pieces of 1 KB per rank, so the
        # field below travels in several all_gathers whose last one is ragged
        from types import SimpleNamespace
        from fibergen_amd.distributed import GlobalViewSolver
        me = SimpleNamespace(nranks=P, group=None, _dist=dist, device=0, GATHER_LIMIT_GB=16.0, GATHER_CHUNK=1024)
        local = (1000.0 * rank + np.arange(6 * 3 * 5 * 7, dtype=np.float64)).reshape(6, 3, 5, 7)
        got = GlobalViewSolver._gather(me, local)
        want = np.concatenate([(1000.0 * r + np.arange(6 * 3 * 5 * 7, dtype=np.float64)).reshape(6, 3, 5, 7) for r in range(P)], axis=1)
        errors = [] if got.shape == want.shape and np.array_equal(got, want) else ["gathered field differs"]
        os.environ["FG_GATHER_LIMIT_GB"] = "1e-9"   # read at call time
        try:
            GlobalViewSolver._gather(me, local)
            errors.append("limit not enforced")
        except RuntimeError:
            pass
        np.savez(a.out + ".%d.npz" % rank, errors=np.array(errors))
        dist.barrier()
        dist.destroy_process_group()
        return
    if a.backend == "fg-shard":
        # calc_effective_properties with the six load cases dealt to the ranks (FG.shard_load_cases)
        from fibergen_amd import FG
        fg = FG(device=0)
        fg.set_xml("""<settings><solver nx="%d" ny="%d" nz="%d"><tol>%g</tol><method>basic</method><mixing_rule>%s</mixing_rule>
          <materials><matrix E="1" nu="0.3" /><inclusion E="10" nu="0.2" /></materials></solver>
          <actions><select_material name="inclusion" /><place_fiber R="0.3" /><calc_effective_properties /></actions>
        </settings>""" % (grid + (a.tol, a.mixing)))
        fg.shard_load_cases(True)
        rc = fg.run()
        np.savez(a.out + ".%d.npz" % rank, rc=rc, C=np.array(fg.get_effective_property()))
        dist.barrier()
        dist.destroy_process_group()
        return
    if a.backend == "fg-slabs":
        # ONE project, its grid cut into x-slabs over the ranks behind the FG interface (FG.decompose_slabs)
        from fibergen_amd import FG
        if a.transport == "rccl":
            os.environ["FG_SLAB_TRANSPORT"] = "rccl"
        fg = FG(device=0)
        fg.set_xml(fg_slabs_project(grid, a.tol, a.method, a.mixing, a.mode))
        fg.decompose_slabs(True)
        rc = fg.run()
        np.savez(a.out + ".%d.npz" % rank, rc=rc, C=np.array(fg.get_effective_property()), eps=fg.get_field("epsilon"),
                 vf=fg.get_volume_fraction("inclusion"), residuals=np.array(fg.get_residuals()))
        dist.barrier()
        dist.destroy_process_group()
        return
    from helpers import two_phase_setup
    from fibergen_amd.distributed import DistributedLSSolver
    if a.mode in ("porous", "heat"):
        # the scalar modes on the slabs: one member per process
        from helpers import sphere_phi
        phi1 = sphere_phi(grid, 0.3)
        s = DistributedLSSolver(*grid, *dims, device=0, transport=a.transport or None)
        s.set_options(mode=a.mode)
        s.set_num_phases(2)
        s.set_phase(0, 1.0, 0.0, s.slab(1 - phi1))
        s.set_phase(1, 12.0, 0.0, s.slab(phi1))
        s.set_options(tol=a.tol, slab_split=a.split)
        failed = s.run(np.array([1.0, -0.5, 0.25]))
        np.savez(a.out + ".%d.npz" % rank, eps=s.get_field("epsilon"), iterations=s.iterations, residuals=np.array(s.residuals),
                 mean_stress=s.mean_stress(), failed=failed, transport=s.transport, mu_0=s.ref_material[0])
        s.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    mats, phis, normals = two_phase_setup(grid, a.mixing)
    try:
        s = DistributedLSSolver(*grid, *dims, device=0, transport=a.transport or None)
    except RuntimeError as e:
        if a.transport == "rccl" and ("ncclCommInitRank" in str(e) or "ncclGetUniqueId" in str(e) or "cannot load librccl" in str(e)):
            # the one failure the GPU suite may skip on: RCCL cannot connect the ranks on this box at all
            print("FG_RCCL_INIT_FAILED: %s" % e, file=sys.stderr, flush=True)
        raise
    nph = 2
    if a.bad_rank >= 0:
        nph = 3
        mats = list(mats) + [mats[1]]
        third = np.zeros_like(phis[0])
        lo = a.bad_rank * (grid[0] // P)
        sl = phis[1][lo:lo + grid[0] // P]
        idx = np.argwhere((sl > 0.2) & (sl < 0.8))[0]
        v = (lo + idx[0], idx[1], idx[2])
        third[v] = 0.1
        phis = [phis[0].copy(), phis[1].copy(), third]
        phis[0][v] -= 0.1
    s.set_num_phases(nph)
    for p in range(nph):
        s.set_phase(p, mats[p][0], mats[p][1], s.slab(phis[p]))
    s.set_normals(s.slab(normals))
    s.set_options(mixing_rule=a.mixing, tol=a.tol, slab_split=a.split, method=a.method)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    S = None
    if a.mixed_bc:
        Pm = np.zeros((6, 6))
        Pm[0, 0] = 1.0
        s.set_bc_projector(Pm)
        s.set_options(bc_tol=1e-8, maxiter=400)
        E = np.array([0.01, 0, 0, 0, 0, 0])
        S = np.zeros(6)
    calls = []
    if a.stop_rank == rank:
        def cb():
            calls.append(1)
            return len(calls) >= a.stop_iter
        s.set_convergence_callback(cb)
    if a.cancel_rank >= 0:
        s.set_options(tol=-1.0, abs_tol=-1.0, maxiter=2000000)   # never converged: only the cancellation ends the run
        if a.cancel_rank == rank:
            import threading
            threading.Timer(1.5, s.cancel).start()
    error = ""
    failed = None
    try:
        if a.step_stop_rank >= 0:
            def step_cb(i):
                calls.append(i)
                return i == 1
            failed = s.run_load_steps(E, S, params=[0.0, 0.5, 1.0],
                                      step_callback=step_cb if a.step_stop_rank == rank else None)
        else:
            failed = s.run(E, S)
    except RuntimeError as e:
        error = str(e)
    if error:
        np.savez(a.out + ".%d.npz" % rank, error=error, transport=s.transport)
    else:
        np.savez(a.out + ".%d.npz" % rank, eps=s.get_field("epsilon"), sigma=s.get_field("sigma"),
                 iterations=s.iterations, residuals=np.array(s.residuals), mean_stress=s.mean_stress(),
                 mean_strain=s.mean_strain(), mu_0=s.ref_material[0], failed=failed, vf=s.volume_fraction(1),
                 transport=s.transport, callback_calls=len(calls), error="")
    s.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
