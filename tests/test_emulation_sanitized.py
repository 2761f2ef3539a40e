"""The host emulation of the kernels' per-thread code (LDS index maps, twiddle tables, the voxeliser's plane cut) once more
under AddressSanitizer + UndefinedBehaviorSanitizer: an LDS index one past a padded row or a signed overflow in an offset
is silent on the GPU (GPU ASan is not available on this pool), on the host build it aborts.  CPU only."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.timeout(900)
def test_emulation_suites_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("no libasan next to gcc")
    env = dict(os.environ, FG_EMU_SANITIZE="1", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_fft_emulation.py"), os.path.join(ROOT, "tests", "test_plane_cut.py")],
                         env=env, capture_output=True, text=True, cwd=ROOT)
    tail = out.stdout[-3000:] + out.stderr[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
