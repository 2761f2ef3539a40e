"""The benchmark RVE generator (fibergen_amd/rve.py): deterministic, and a rank's x-slab generated on
its own equals the slab of the full field (bench.py --gpus N generates per rank)."""
import numpy as np

from fibergen_amd.rve import bench_rve, bench_rve_parameters


def test_slab_of_the_rve_equals_the_full_field():
    phi, nrm, par = bench_rve(64, "laminate")
    assert par == bench_rve_parameters(64)
    again, nrm2, _ = bench_rve(64, "laminate")
    assert np.array_equal(phi, again) and np.array_equal(nrm, nrm2)          # thread schedule does not matter
    for lo, hi in ((0, 8), (24, 40), (56, 64)):
        p, n, _ = bench_rve(64, "laminate", x_range=(lo, hi))
        assert p.shape == (hi - lo, 64, 64)
        assert np.array_equal(p, phi[lo:hi]) and np.array_equal(n, nrm[:, lo:hi])
    assert 0.05 < phi.mean() < 0.25 and phi.min() == 0.0 and phi.max() == 1.0
    mixed = (phi > 0) & (phi < 1)
    assert np.abs(np.linalg.norm(nrm[:, mixed], axis=0) - 1).max() < 1e-12       # unit normals at every interface voxel
