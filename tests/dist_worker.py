"""Worker for the multi-rank tests: run under torch.distributed.run with 2+ ranks.
    --backend fake : NumPy slab backend over gloo (CPU suite)
    --backend hip  : the HIP slab kernels; several ranks may share one GPU, bytes are then
                     staged through the host by gloo (GPU suite on a 1-GPU box)
Each rank writes its slab of the converged strain + scalars to --out.<rank>.npz."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="fake")
    ap.add_argument("--grid", default="8,8,6")
    ap.add_argument("--dims", default="1,1,1")
    ap.add_argument("--mixing", default="voigt")
    ap.add_argument("--tol", type=float, default=1e-8)
    ap.add_argument("--mixed-bc", type=int, default=0)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch  # before the HIP library: one shared runtime
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    grid = tuple(int(v) for v in a.grid.split(","))
    dims = tuple(float(v) for v in a.dims.split(","))
    from helpers import two_phase_setup
    from fibergen_amd.distributed import DistributedLSSolver, HipSlabBackend
    mats, phis, normals = two_phase_setup(grid, a.mixing)
    if a.backend == "fake":
        from fake_slab_backend import FakeSlabBackend
        be = FakeSlabBackend(*grid, *dims, rank, P, a.mixing)
    else:
        be = HipSlabBackend(*grid, *dims, rank, P, device=0)
    s = DistributedLSSolver(*grid, *dims, backend=be)
    s.set_num_phases(2)
    for p in range(2):
        s.set_phase(p, mats[p][0], mats[p][1], s.slab(phis[p]))
    s.set_normals(s.slab(normals))
    s.set_options(mixing_rule=a.mixing, tol=a.tol)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    S = None
    if a.mixed_bc:
        Pm = np.zeros((6, 6))
        Pm[0, 0] = 1.0
        s.set_bc_projector(Pm)
        s.set_options(bc_tol=1e-8, maxiter=400)
        E = np.array([0.01, 0, 0, 0, 0, 0])
        S = np.zeros(6)
    failed = s.run(E, S)
    np.savez(a.out + ".%d.npz" % rank, eps=s.get_field("epsilon"), sigma=s.get_field("sigma"),
             iterations=s.iterations, residuals=np.array(s.residuals), mean_stress=s.mean_stress(),
             mean_strain=s.mean_strain(), mu_0=s.mu_0, failed=failed, vf=s.volume_fraction(1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
