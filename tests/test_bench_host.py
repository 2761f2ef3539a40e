"""Host-side pieces of bench.py that need no GPU: the algorithmic-byte table the roofline figures are priced with,
the kernel-name matching of the live counter passes, the CPU stand-in's child run (oracle/cpu_loop.py) and the limits the
baseline reports."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_follow_survey_8d():
    n = (256, 256, 256)
    N, F = 256 ** 3, 256 * 256 * 129
    ab = bench.algorithmic_bytes(n, 2)
    assert ab["stress"] == 104 * N and ab["div"] == 72 * N and ab["eps_norm"] == 72 * N   # S 13*8, div 9*8, eps 9*8
    assert ab["g0"] == 96 * F == 811597824                                                  # the K4 figure of the bench line
    assert ab["r2c_z"] == ab["c2c_y_fwd"] == ab["c2r_z"] == 96 * F                           # 3 components, read + write, complex128
    # A_stage = 632 B/voxel for the reference's pass structure (FFT counted per real voxel: 48 B per pass)
    assert bench.A_STAGE_BYTES_PER_VOXEL == 104 + 72 + 144 + 48 + 144 + 72 + 48
    assert bench.A_MIN_BYTES_PER_VOXEL == 80 + 48 + 72 + 192


def test_cpu_limits_and_throttle_counters_are_readable():
    lim = bench.cpu_limits()
    assert lim["affinity_cpus"] is None or lim["affinity_cpus"] >= 1
    assert lim["cgroup_quota_cpus"] is None or lim["cgroup_quota_cpus"] > 0
    thr = bench.cpu_throttled()
    assert thr is None or (thr[0] >= 0 and thr[1] >= 0.0)


def test_cpu_loop_child_run(tmp_path):
    """oracle/cpu_loop.py: the process bench.py starts per thread count -- pinned OpenMP threads, own FFT, both loop orders give the
    same norms (the JSON carries what was run)."""
    from fibergen_amd.rve import bench_rve
    phi, normals, _ = bench_rve(32, "laminate")
    np.save(tmp_path / "phi.npy", phi)
    np.save(tmp_path / "nrm.npy", normals)
    out = {}
    for loops in ("reference", "contiguous"):
        env = dict(os.environ, OMP_NUM_THREADS="2", OMP_PROC_BIND="spread", OMP_PLACES="cores", FG_REF_NATIVE_BUILT="0")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_loop.py"), "--grid", "32", "--mixing", "laminate",
                            "--phi", str(tmp_path / "phi.npy"), "--normals", str(tmp_path / "nrm.npy"), "--threads", "2",
                            "--max-passes", "2", "--max-seconds", "0.1", "--loops", loops],
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        out[loops] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
        assert out[loops]["fft"] == "own" and out[loops]["loops"] == loops and out[loops]["passes"] == 2 and out[loops]["it_s"] > 0
