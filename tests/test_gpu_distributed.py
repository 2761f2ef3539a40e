"""GPU tests of the slab driver with ONE MEMBER PER PROCESS (torch.distributed over gloo; the test box has one GPU, so
the ranks share it and the callback transport stages the exchanged bytes through the host).  The processes run the
collective entry points of the C ABI exactly as an 8-GPU RCCL job does: separate solver objects, the exchange plan,
all-reduced norms, identical stop decisions on every rank."""
import os

import numpy as np
import pytest

from helpers import make_oracle, rel_err
from test_distributed_cpu import launch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nproc,grid,mixing,split", [
    (2, "8,16,128", "voigt", 1),     # displacement loop (tiled sweep with halo planes), all-to-all per component
    (2, "8,16,128", "voigt", 0),     # ... one exchange for the three components
    (2, "8,16,128", "laminate", 1),  # displacement loop with the interface correction (dense planes to the neighbours)
    (4, "16,16,128", "laminate", 0),
    (2, "16,16,16", "voigt", 1),     # strain-state pipeline
    (2, "32,16,64", "laminate", 0),
    (4, "16,8,16", "voigt", 1),
    (2, "12,10,6", "laminate", 1),
])
def test_hip_slabs_match_oracle(tmp_path, nproc, grid, mixing, split):
    g = tuple(int(v) for v in grid.split(","))
    res = launch(nproc, str(tmp_path / "r"), "--backend", "hip", "--grid", grid, "--mixing", mixing, "--dims", "1,2,1.5",
                 "--split", str(split))
    o = make_oracle(g, (1.0, 2.0, 1.5), mixing, tol=1e-8)
    assert o.run([1.0, 0, 0, 0, 0, 0.5]) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    sig = np.concatenate([r["sigma"] for r in res], axis=1)
    assert all(int(r["iterations"]) == o.iterations for r in res)
    assert all(str(r["transport"]) == "callback" for r in res)
    assert rel_err(eps, o.eps) < 1e-9
    assert rel_err(sig, o.get_field("sigma")) < 1e-9
    for r in res:
        assert np.array_equal(r["residuals"], res[0]["residuals"])
        assert np.abs(r["residuals"] - np.array(o.residuals)).max() < 1e-11
        assert rel_err(r["mean_stress"], o.mean_stress()) < 1e-10
        assert float(r["mu_0"]) == pytest.approx(o.mu_0, rel=1e-14)
        assert float(r["vf"]) == pytest.approx(float(o.phis[1].mean()), rel=1e-13)


def test_hip_slabs_mixed_bc(tmp_path):
    res = launch(2, str(tmp_path / "m"), "--backend", "hip", "--grid", "16,16,16", "--mixed-bc", "1", "--tol", "1e-9")
    o = make_oracle((16, 16, 16), tol=1e-9, bc_tol=1e-8, maxiter=400)
    P = np.zeros((6, 6))
    P[0, 0] = 1
    assert o.run([0.01, 0, 0, 0, 0, 0], S0=np.zeros(6), P=P) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert int(res[0]["iterations"]) == o.iterations
    assert rel_err(eps, o.eps) < 1e-8


@pytest.mark.parametrize("nproc", [2, 4])
def test_fg_load_cases_sharded_over_ranks(tmp_path, nproc):
    """FG.shard_load_cases: the six unit experiments of calc_effective_properties dealt to the ranks (no data-path
    collective, only the 6 x 6 means are gathered); every rank ends with the stiffness one process computes alone."""
    from fibergen_amd import FG
    res = launch(nproc, str(tmp_path / "s"), "--backend", "fg-shard", "--grid", "16,16,16", "--tol", "1e-8")
    fg = FG()
    fg.set_xml("""<settings><solver n="16"><tol>1e-8</tol><method>basic</method><mixing_rule>voigt</mixing_rule>
      <materials><matrix E="1" nu="0.3" /><inclusion E="10" nu="0.2" /></materials></solver>
      <actions><select_material name="inclusion" /><place_fiber R="0.3" /><calc_effective_properties /></actions>
    </settings>""")
    assert fg.run() == 0
    C = np.array(fg.get_effective_property())
    for r in res:
        assert int(r["rc"]) == 0
        assert np.array_equal(r["C"], res[0]["C"])
        assert rel_err(r["C"], C) < 1e-12


def launch_rccl(nproc, out, *args):
    """launch(..., --transport rccl).  Skips ONLY when RCCL cannot connect the ranks on this box at all (no `lo`, sockets
    forbidden): the worker prints FG_RCCL_INIT_FAILED when ncclGetUniqueId / ncclCommInitRank fail.  Errors of the exchanges
    or the all-reduces (wrong counts, peers, group nesting) and wrong RESULTS fail the test."""
    import subprocess
    try:
        return launch(nproc, out, *args, "--transport", "rccl")
    except subprocess.CalledProcessError as e:
        err = (e.stderr or b"").decode(errors="replace")
        if "FG_RCCL_INIT_FAILED" in err:
            line = [ln for ln in err.splitlines() if "FG_RCCL_INIT_FAILED" in ln][-1]
            pytest.skip("RCCL could not connect the ranks over loop-back sockets here: " + line[:200])
        raise AssertionError("worker failed:\n" + err[-3000:]) from e


@pytest.mark.parametrize("nproc,grid,mixing,split", [
    (2, "8,16,128", "voigt", 1),      # displacement loop, one all-to-all per component on the second stream
    (2, "8,16,128", "laminate", 0),   # + the interface correction's dense planes, three components in one exchange
    (4, "16,16,128", "voigt", 1),     # four ranks: every rank sends to three peers per all-to-all
    (4, "16,16,128", "laminate", 1),
    (8, "32,16,128", "laminate", 0),  # eight ranks as on a full node: 7 peers per all-to-all, slabs of 4 planes
    (2, "12,10,6", "laminate", 1),    # strain-state pipeline, remapped all-to-all layout
])
def test_rccl_transport_between_ranks_on_one_gpu(tmp_path, nproc, grid, mixing, split):
    """The RCCL transport with MORE THAN ONE RANK on the one GPU of the test box: every process poses as a host of its own
    (NCCL_HOSTID), so RCCL connects the ranks through its socket transport on the loop-back interface instead of refusing the
    duplicate device.  ncclCommInitRank over 2 / 4 ranks, grouped ncclSend / ncclRecv between DIFFERENT ranks (all-to-all
    blocks, halo planes to the left and right neighbour), ncclAllReduce of the norms -- the calls, counts, peers, ordering
    and stream choreography of the multi-GPU run; only the wire differs (sockets instead of xGMI)."""
    g = tuple(int(v) for v in grid.split(","))
    res = launch_rccl(nproc, str(tmp_path / "q"), "--backend", "hip", "--grid", grid, "--mixing", mixing, "--dims", "1,2,1.5",
                      "--split", str(split))
    o = make_oracle(g, (1.0, 2.0, 1.5), mixing, tol=1e-8)
    assert o.run([1.0, 0, 0, 0, 0, 0.5]) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert all(str(r["transport"]) == "rccl" for r in res)
    assert all(int(r["iterations"]) == o.iterations for r in res)
    assert rel_err(eps, o.eps) < 1e-9
    for r in res:
        assert np.array_equal(r["residuals"], res[0]["residuals"])      # all-reduced norms: identical on every rank
        assert np.abs(r["residuals"] - np.array(o.residuals)).max() < 1e-11
        assert rel_err(r["mean_stress"], o.mean_stress()) < 1e-10


def test_rccl_ranks_mixed_bc(tmp_path):
    res = launch_rccl(2, str(tmp_path / "qm"), "--backend", "hip", "--grid", "8,16,128", "--mixed-bc", "1", "--tol", "1e-9")
    o = make_oracle((8, 16, 128), tol=1e-9, bc_tol=1e-8, maxiter=400)
    P = np.zeros((6, 6))
    P[0, 0] = 1
    assert o.run([0.01, 0, 0, 0, 0, 0], S0=np.zeros(6), P=P) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert str(res[0]["transport"]) == "rccl" and int(res[0]["iterations"]) == o.iterations
    assert rel_err(eps, o.eps) < 1e-8


@pytest.mark.parametrize("transport", ["callback", "rccl"])
def test_loadstep_callback_on_one_rank_stops_every_rank(tmp_path, transport):
    """a load-step callback that exists on ONE rank only answers for all of them (its break request is voted over
    the ranks after every step); before, that rank returned alone and the others hung in the next step's collectives."""
    args = ("--backend", "hip", "--grid", "8,16,128", "--step-stop-rank", "1")
    res = (launch_rccl(2, str(tmp_path / "l"), *args) if transport == "rccl" else launch(2, str(tmp_path / "l"), *args))
    assert [bool(r["failed"]) for r in res] == [True, True]          # "Loadstep callback break request" on both ranks
    assert [int(r["callback_calls"]) for r in res] == [0, 2]         # steps 0 and 1 were seen by rank 1 only
    assert np.array_equal(res[0]["residuals"], res[1]["residuals"])


# ---- collective stop and error decisions (round 3): a rank that alone wants to stop, or alone sees a device-side error, must
#      take every other rank with it in the same pass -- the others have already enqueued the next pass's exchanges
@pytest.mark.parametrize("transport", ["callback", "rccl"])
@pytest.mark.parametrize("grid,mixing", [("8,16,128", "voigt"), ("16,16,16", "voigt")])   # displacement loop / strain-state pipeline
def test_callback_on_one_rank_stops_every_rank(tmp_path, transport, grid, mixing):
    args = ("--backend", "hip", "--grid", grid, "--mixing", mixing, "--stop-rank", "1", "--stop-iter", "3")
    res = (launch_rccl(2, str(tmp_path / "c"), *args) if transport == "rccl" else launch(2, str(tmp_path / "c"), *args))
    assert [int(r["iterations"]) for r in res] == [3, 3]
    assert [int(r["callback_calls"]) for r in res] == [0, 3]
    assert np.array_equal(res[0]["residuals"], res[1]["residuals"]) and len(res[0]["residuals"]) == 3
    assert not bool(res[0]["failed"]) and not bool(res[1]["failed"])
    # the run that was stopped equals the first three passes of an undisturbed one
    g = tuple(int(v) for v in grid.split(","))
    o = make_oracle(g, (1.0, 1.0, 1.0), mixing, tol=1e-8, maxiter=3)
    o.run([1.0, 0, 0, 0, 0, 0.5])
    assert rel_err(np.concatenate([r["eps"] for r in res], axis=1), o.eps) < 1e-9


@pytest.mark.parametrize("transport", ["callback", "rccl"])
@pytest.mark.parametrize("grid", ["8,16,128", "32,16,64"])
def test_device_error_on_one_rank_raises_on_every_rank(tmp_path, transport, grid):
    """laminate mixing with a three-phase voxel in rank 1's slab only: both ranks must raise, in the same pass"""
    args = ("--backend", "hip", "--grid", grid, "--mixing", "laminate", "--bad-rank", "1")
    res = (launch_rccl(2, str(tmp_path / "e"), *args) if transport == "rccl" else launch(2, str(tmp_path / "e"), *args))
    for r in res:
        assert "laminate mixing rule supports only two phase mixtures" in str(r["error"])
    assert "another rank" in str(res[0]["error"]) and "another rank" not in str(res[1]["error"])


@pytest.mark.parametrize("loopback", [0, 1])
def test_lone_slab_on_rccl_transport(loopback):
    """fg_create_slab(..., 0, 1) + fg_slab_connect_rccl: with loop-back the slab sends to itself through RCCL; without it
    nothing is exchanged, but the sums still reach the host over the exchange stream, which must follow the sweep
    (the copies were unordered)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_loopback_worker.py"), str(loopback)],
                       capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and "FG_RCCL_INIT_FAILED" in p.stderr:
        pytest.skip("RCCL cannot be initialised here: " + p.stderr.strip().splitlines()[-1][:200])
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout + p.stderr


@pytest.mark.parametrize("transport", ["callback", "rccl"])
@pytest.mark.parametrize("nproc,grid,mixing,split", [(2, "8,16,128", "voigt", 0), (2, "8,16,128", "laminate", 1),
                                                     (4, "16,16,128", "laminate", 0),
                                                     (2, "16,16,16", "laminate", 1)])    # strain-space form
def test_cg_on_slabs_one_rank_per_process(tmp_path, transport, nproc, grid, mixing, split):
    """method = cg over the ranks: the inner products are all-reduced on the device, iteration counts and residual
    histories equal the oracle's CG on every rank"""
    g = tuple(int(v) for v in grid.split(","))
    args = ("--backend", "hip", "--grid", grid, "--mixing", mixing, "--dims", "1,2,1.5", "--split", str(split), "--method", "cg",
            "--tol", "1e-9")
    res = launch_rccl(nproc, str(tmp_path / "g"), *args) if transport == "rccl" else launch(nproc, str(tmp_path / "g"), *args)
    o = make_oracle(g, (1.0, 2.0, 1.5), mixing, tol=1e-9)
    assert o.run_cg([1.0, 0, 0, 0, 0, 0.5]) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert all(int(r["iterations"]) == o.iterations for r in res)
    assert rel_err(eps, o.eps) < 1e-8
    for r in res:
        assert np.array_equal(r["residuals"], res[0]["residuals"])
        assert np.abs(r["residuals"] - np.array(o.residuals)).max() < 1e-9
        assert rel_err(r["mean_stress"], o.mean_stress()) < 1e-9


@pytest.mark.parametrize("nproc,mixing,method", [(2, "voigt", "basic"), (4, "laminate", "basic"), (2, "laminate", "cg")])
def test_rccl_ranks_at_128_cubed_equal_single_gpu_solver(tmp_path, nproc, mixing, method):
    """BASELINE config 2's size through real RCCL ranks (sharing the GPU, loop-back sockets): 128^3, slabs of 64 / 32 planes,
    blocks of 0.8-1.6 MB per peer and component -- against the single-GPU solver on the same problem (which the oracle checks
    at this size in tests/test_gpu_fullsize_oracle.py)."""
    from helpers import make_gpu_solver
    grid = (128, 128, 128)
    res = launch_rccl(nproc, str(tmp_path / "b"), "--backend", "hip", "--grid", "128,128,128", "--mixing", mixing, "--tol", "1e-5",
                      "--method", method)
    s = make_gpu_solver(grid, mixing=mixing, tol=1e-5, method=method)
    assert s.run([1.0, 0, 0, 0, 0, 0.5]) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert all(str(r["transport"]) == "rccl" and int(r["iterations"]) == s.iterations for r in res)
    assert np.abs(res[0]["residuals"] - np.array(s.residuals)).max() < 1e-11
    assert rel_err(eps, s.get_field("epsilon")) < 1e-10
    assert rel_err(res[0]["mean_stress"], s.mean_stress()) < 1e-11
    s.close()


@pytest.mark.parametrize("transport", ["callback", "rccl"])
@pytest.mark.parametrize("method", ["basic", "cg"])
def test_cancel_on_one_rank_stops_every_rank(tmp_path, transport, method):
    """fg_cancel from another thread on rank 1 only, in a run that would never stop (tol 0): the request travels with the flag
    word of the next reduction, every rank leaves the loop in the same pass and reports failure like the reference's
    cancelled run (F:21206)."""
    args = ("--backend", "hip", "--grid", "8,16,128", "--cancel-rank", "1", "--method", method)
    res = (launch_rccl(2, str(tmp_path / "x"), *args) if transport == "rccl" else launch(2, str(tmp_path / "x"), *args))
    assert bool(res[0]["failed"]) and bool(res[1]["failed"])
    assert int(res[0]["iterations"]) == int(res[1]["iterations"]) > 3
    assert np.array_equal(res[0]["residuals"], res[1]["residuals"])


@pytest.mark.parametrize("nproc,grid,mixing,method,transport,mode", [
    (2, "8,16,128", "voigt", "cg", "callback", "elasticity"),
    (2, "8,16,128", "laminate", "basic", "rccl", "elasticity"),
    (4, "16,16,16", "voigt", "cg", "callback", "elasticity"),
    (2, "8,16,128", "voigt", "basic", "rccl", "porous"),
    (2, "12,10,6", "voigt", "cg", "rccl", "porous"),
    (2, "12,10,6", "voigt", "cg", "callback", "viscosity"),
])
def test_fg_project_on_slabs(tmp_path, nproc, grid, mixing, method, transport, mode):
    """FG.decompose_slabs: the project interface on top of the slab driver -- one XML project, its voxel grid cut into
    x-slabs over the ranks (north_star: 'behind the same ... XML project interface ... slab-decomposed across the GPUs').
    Every rank obtains the effective property, gathered fields and scalars one process computes alone."""
    from dist_worker import fg_slabs_project
    from fibergen_amd import FG
    g = [int(v) for v in grid.split(",")]
    args = ("--backend", "fg-slabs", "--grid", grid, "--mixing", mixing, "--method", method, "--tol", "1e-8", "--mode", mode)
    res = launch_rccl(nproc, str(tmp_path / "f"), *args) if transport == "rccl" else launch(nproc, str(tmp_path / "f"), *args)
    fg = FG()
    fg.set_xml(fg_slabs_project(g, 1e-8, method, mixing, mode))
    assert fg.run() == 0
    C = np.array(fg.get_effective_property())
    ncomp = 3 if mode == "porous" else 6
    for r in res:
        assert int(r["rc"]) == 0
        assert np.array_equal(r["C"], res[0]["C"]) and rel_err(r["C"], C) < 1e-9
        assert r["eps"].shape == (ncomp, g[0], g[1], g[2]) and rel_err(r["eps"], fg.get_field("epsilon")) < 1e-8
        assert float(r["vf"]) == pytest.approx(fg.get_volume_fraction("inclusion"), rel=1e-13)
        assert len(r["residuals"]) == len(fg.get_residuals())


@pytest.mark.parametrize("transport", ["callback", "rccl"])
@pytest.mark.parametrize("nproc,grid,split", [(2, "8,16,128", 0), (4, "16,16,128", 1),
                                              (2, "12,10,7", 0)])   # untiled sweep, separate x pass on the y-slab
def test_porous_mode_on_slabs_one_rank_per_process(tmp_path, transport, nproc, grid, split):
    """mode = porous (BASELINE config 5's physics) cut into slabs: potential with halo planes, one-component all-to-all"""
    from helpers import sphere_phi
    from oracle.scalar_oracle import ScalarOracle
    g = tuple(int(v) for v in grid.split(","))
    args = ("--backend", "hip", "--grid", grid, "--mode", "porous", "--dims", "1,2,1.5", "--split", str(split), "--tol", "1e-9")
    res = launch_rccl(nproc, str(tmp_path / "p"), *args) if transport == "rccl" else launch(nproc, str(tmp_path / "p"), *args)
    phi1 = sphere_phi(g, 0.3)
    o = ScalarOracle(*g, mus=[1.0, 12.0], phis=[1 - phi1, phi1], dx=1.0, dy=2.0, dz=1.5, tol=1e-9)
    assert o.run(np.array([1.0, -0.5, 0.25])) is False
    eps = np.concatenate([r["eps"] for r in res], axis=1)
    assert all(int(r["iterations"]) == o.iterations for r in res)
    assert rel_err(eps, o.eps) < 1e-10
    for r in res:
        assert np.array_equal(r["residuals"], res[0]["residuals"])
        assert np.abs(r["residuals"] - np.array(o.residuals)).max() < 1e-11
        assert rel_err(r["mean_stress"][:3], o.mean_stress()) < 1e-11
        assert float(r["mu_0"]) == o.mu_0


@pytest.mark.parametrize("backend,launcher", [("nccl-one-gpu", "self"), ("gloo", "self"), ("nccl-one-gpu", "torchrun")])
def test_bench_line_for_two_ranks(backend, launcher):
    """`bench.py --gpus 2` (what the driver starts per N -- self-launched and under torch.distributed.run, both ranks on the one GPU of the box): the slab
    section runs, the line carries the fields the contract and north_star's table ask for -- value over ALL ranks,
    the same node's single-GPU rate, the kernels of rank 0's slab, the exchange times, the other grid through the same driver."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", backend, "--size", "128", "--steps", "4",
           "--warmup", "2", "--repeats", "3", "--no-cpu-baseline", "--also-slab", "64:laminate", "--slab-timeout", "120"]
    if launcher == "torchrun":   # the driver's command line for N > 1
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + cmd[1:]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["warmup"] == 2 and line["unit"] == "it/s"
    assert line["value"] and line["value"] > 0 and abs(line["ms_per_step"] * line["value"] - 1e3) < 1e-6 * 1e3
    assert line["scaling"] == "strong" and line["config"]["grid"] == [128, 128, 128]
    assert line["single_gpu_it_s"] > 0 and line["speedup_over_single_gpu"] > 0 and line["replicas"]["value"] > 0
    assert set(line["kernels"]) >= {"u_eps_stress_div", "xfft_g0_xifft"} and line["roofline"]["frac"] > 0
    assert line["alltoall_ms"] >= 0 and line["distinct_devices"] == 1
    assert line["also_slab"]["64^3 laminate"]["it_s"] > 0
