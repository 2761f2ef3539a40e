"""examples/c_abi_example.c: the drop-in boundary used from plain C99 (the header is C, the library needs nothing but itself).
CPU: compiles with -pedantic, links, and fails loudly without a GPU (no CPU fallback); GPU: runs and agrees with the Python
wrapper on the same cell."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "c_abi_example")
    libdir = os.path.join(ROOT, "fibergen_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_example.c"), "-L", libdir, "-lfibergen_amd",
                           "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    return exe


def test_c_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    if not os.path.exists(os.path.join(ROOT, "fibergen_amd", "libfibergen_amd.so")):
        pytest.skip("library not built")
    exe = _build(tmp_path)
    out = subprocess.run([exe, "8"], capture_output=True, text=True, timeout=300)
    if out.returncode == 0:          # a GPU is present after all
        assert "mean stress:" in out.stdout
    else:
        assert out.returncode == 4 and "fg_create: HIP error" in out.stderr, (out.returncode, out.stderr)


@pytest.mark.gpu
def test_c_example_agrees_with_the_python_wrapper(tmp_path):
    from fibergen_amd import LSSolver
    exe = _build(tmp_path)
    n = 32
    out = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("mean stress:")][0]
    sig_c = np.array([float(v) for v in line.split(":")[1].split()])
    x = (np.arange(n) + 0.5) / n - 0.5
    r = np.sqrt(x[:, None, None] ** 2 + x[None, :, None] ** 2 + x[None, None, :] ** 2)
    phi1 = (r < 0.3).astype(float)
    s = LSSolver(n, n, n)
    s.set_num_phases(2)
    s.set_phase(0, 1.0 / 2.6, 0.3 / (1.3 * 0.4), 1 - phi1)
    s.set_phase(1, 10.0 / 2.4, 2.0 / (1.2 * 0.6), phi1)
    s.set_options(tol=1e-8, method="cg")
    assert s.run(np.array([0.01, 0, 0, 0, 0, 0])) is False
    ref = np.asarray(s.mean_stress())
    assert np.abs(sig_c - ref).max() < 1e-9 * np.abs(ref).max()   # (printed with 10 digits)
    s.close()
