"""Single-process worker: PyTorch imported FIRST (as bench.py does under torchrun), then a lone slab with the library's own
RCCL communicator in loop-back mode -- the RCCL the library dlopens is then the one PyTorch ships.  Prints 'OK <librccl>'."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    torch.cuda.init()
    from helpers import make_oracle, rel_err, two_phase_setup
    from fibergen_amd.distributed import SlabMember, rccl_unique_id
    grid = (8, 16, 128)
    mats, phis, normals = two_phase_setup(grid, "voigt")
    loopback = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    m = SlabMember(*grid, rank=0, nranks=1)
    try:
        m.connect_rccl(rccl_unique_id())
    except RuntimeError as e:
        print("FG_RCCL_INIT_FAILED: %s" % e, file=sys.stderr, flush=True)
        raise
    m.set_num_phases(2)
    for p in range(2):
        m.set_phase(p, mats[p][0], mats[p][1], phis[p])
    m.set_options(tol=1e-8, slab_loopback=loopback, slab_split=1)
    o = make_oracle(grid, tol=1e-8)
    E = np.array([1.0, 0, 0, 0, 0, 0.5])
    assert o.run(E) is False and m.run(E) is False
    assert m.iterations == o.iterations and rel_err(m.get_field("epsilon"), o.eps) < 1e-9
    assert np.abs(np.array(m.residuals) - np.array(o.residuals)).max() < 1e-11
    assert rel_err(m.mean_stress(), o.mean_stress()) < 1e-10 and rel_err(m.mean_strain(), o.eps.mean(axis=(1, 2, 3))) < 1e-12
    libs = [ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln]
    print("OK", sorted(set(libs)))
    m.close()


if __name__ == "__main__":
    main()
