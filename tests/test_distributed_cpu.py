"""The N > 1 path on CPU: fibergen_amd.distributed.DistributedLSSolver driven with a NumPy
slab backend over gloo, world_size 2 and 4, against the single-process oracle.  Checks the
exchange pattern (halo planes, all-to-all blocks), the rank-ordered reductions, the stop rule
and the mixed-BC mean correction across ranks."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import make_oracle, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(nproc, out, *args, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = str(29500 + (os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.join(ROOT, "tests", "dist_worker.py"),
           "--out", out, *args]
    subprocess.run(cmd, check=True, env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    return [np.load(out + ".%d.npz" % r) for r in range(nproc)]


@pytest.mark.parametrize("nproc,grid,mixing", [(2, "8,8,6", "voigt"), (2, "8,6,5", "laminate"), (4, "8,8,4", "voigt")])
def test_slab_driver_matches_single_process_oracle(tmp_path, nproc, grid, mixing):
    g = tuple(int(v) for v in grid.split(","))
    res = launch(nproc, str(tmp_path / "r"), "--backend", "fake", "--grid", grid, "--mixing", mixing, "--dims", "1,2,1.5")
    o = make_oracle(g, (1.0, 2.0, 1.5), mixing, tol=1e-8)
    assert o.run([1.0, 0, 0, 0, 0, 0.5]) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert all(int(r["iterations"]) == o.iterations for r in res)
    assert rel_err(eps, o.eps) < 1e-11
    for r in res:
        assert np.array_equal(r["residuals"], res[0]["residuals"])      # identical decisions on every rank
        assert np.abs(r["residuals"] - np.array(o.residuals)).max() < 1e-12
        assert rel_err(r["mean_stress"], o.mean_stress()) < 1e-11
        assert float(r["mu_0"]) == pytest.approx(o.mu_0, rel=1e-14)
        assert float(r["vf"]) == pytest.approx(float(o.phis[1].mean()), rel=1e-13)


def test_slab_driver_mixed_bc(tmp_path):
    res = launch(2, str(tmp_path / "m"), "--backend", "fake", "--grid", "8,8,8", "--mixed-bc", "1", "--tol", "1e-9")
    o = make_oracle((8, 8, 8), tol=1e-9, bc_tol=1e-8, maxiter=400)
    P = np.zeros((6, 6))
    P[0, 0] = 1
    assert o.run([0.01, 0, 0, 0, 0, 0], S0=np.zeros(6), P=P) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert int(res[0]["iterations"]) == o.iterations
    assert rel_err(eps, o.eps) < 1e-9
    assert np.abs(res[0]["mean_stress"][1:]).max() < 1e-7
