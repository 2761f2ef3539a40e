"""Host emulation of the FFT kernels' per-thread phase code (the very functions the HIP
kernels call between __syncthreads) against numpy.fft.  CPU only; validates index math,
twiddles, LDS padding map and the FFTW r2c/c2r conventions without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from helpers import emulation_build_flags

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dp = ctypes.POINTER(ctypes.c_double)


def P(a):
    return a.ctypes.data_as(dp)


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("emu") / "emu_fft.so")
    subprocess.check_call(["g++"] + emulation_build_flags() + ["-o", out, os.path.join(ROOT, "tests", "emulate", "emu_fft.cpp")])
    return ctypes.CDLL(out)


@pytest.mark.parametrize("N", [8, 16, 32, 64, 128, 256, 512, 1024])
@pytest.mark.parametrize("d", [-1, 1])
def test_strided_c2c(emu, N, d):
    rng = np.random.default_rng(N)
    ncols, nouter = 11, 2  # ragged last tile
    x = rng.standard_normal((nouter, N, ncols)) + 1j * rng.standard_normal((nouter, N, ncols))
    y = x.copy()
    assert emu.emu_strided(N, d, P(y.view(np.float64)), ncols, nouter, ctypes.c_double(0.5)) == 0
    ref = (np.fft.fft(x, axis=1) if d < 0 else np.fft.ifft(x, axis=1) * N) * 0.5
    assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-14


@pytest.mark.parametrize("nz", [16, 32, 64, 128, 256, 512, 1024, 2048])
def test_r2c_c2r(emu, nz):
    rng = np.random.default_rng(nz)
    nrows = 5
    nzc = nz // 2 + 1
    x = rng.standard_normal((nrows, nz))
    buf = np.full((nrows, 2 * nzc), np.nan)
    buf[:, :nz] = x
    assert emu.emu_r2c(nz, P(buf), ctypes.c_long(nrows)) == 0
    ref = np.fft.rfft(x, axis=1)
    assert np.abs(buf.view(np.complex128) - ref).max() / np.abs(ref).max() < 1e-14
    # non-Hermitian input: imaginary parts of DC / Nyquist must be ignored like FFTW's c2r
    X = rng.standard_normal((nrows, nzc)) + 1j * rng.standard_normal((nrows, nzc))
    buf = X.copy().view(np.float64).copy()
    assert emu.emu_c2r(nz, P(buf), ctypes.c_long(nrows)) == 0
    ref = np.fft.irfft(X, n=nz, axis=1) * nz
    assert np.abs(buf[:, :nz] - ref).max() / np.abs(ref).max() < 1e-14


@pytest.mark.parametrize("n", [1, 2, 3, 5, 10, 11, 33, 41])
def test_generic_dft(emu, n):
    rng = np.random.default_rng(n)
    ncols, nouter = 3, 2
    x = rng.standard_normal((nouter, n, ncols)) + 1j * rng.standard_normal((nouter, n, ncols))
    for d in (-1, 1):
        y = np.zeros_like(x)
        emu.emu_dft_strided(P(x.view(np.float64)), P(y.view(np.float64)), n, ncols, nouter, d, ctypes.c_double(1.0))
        ref = np.fft.fft(x, axis=1) if d < 0 else np.fft.ifft(x, axis=1) * n
        assert np.abs(y - ref).max() < 1e-12 * max(1, np.abs(ref).max())
    nrows = 4
    nzc = n // 2 + 1
    xr = rng.standard_normal((nrows, n))
    src = np.zeros((nrows, 2 * nzc))
    src[:, :n] = xr
    dst = np.zeros((nrows, 2 * nzc))
    emu.emu_r2c_generic(P(src), P(dst), n, ctypes.c_long(nrows))
    assert np.abs(dst.view(np.complex128) - np.fft.rfft(xr, axis=1)).max() < 1e-12
    X = rng.standard_normal((nrows, nzc)) + 1j * rng.standard_normal((nrows, nzc))
    src = X.copy().view(np.float64).copy()
    dst = np.zeros((nrows, 2 * nzc))
    emu.emu_c2r_generic(P(src), P(dst), n, ctypes.c_long(nrows))
    ref = np.fft.irfft(X, n=n, axis=1) * n
    assert np.abs(dst[:, :n] - ref).max() < 1e-12 * max(1, np.abs(ref).max())


@pytest.mark.parametrize("N", [8, 32, 64, 128, 256, 512, 1024])
def test_fused_x_pass(emu, N):
    """The fused kernel of the hot loop (x-FFT, 1/N, Green operator with the transformed axis' factors rebuilt from
    e^{i pi kx/N}, inverse x-FFT) run thread by thread on the host against numpy: G0OperatorFourierStaggeredGeneral
    F:19834-19927 per frequency, padding columns (kz >= nzf) passed through, zero frequency set to zero."""
    rng = np.random.default_rng(N)
    ny, nzc, nzf = 3, 8, 6           # ragged: 24 columns, the last two kz of every row are padding
    ncols = ny * nzc
    x = rng.standard_normal((3, N, ncols)) + 1j * rng.standard_normal((3, N, ncols))
    h = [1.0 / (2 * N), 2.0 / (2 * ny), 0.5 / (2 * 10)]

    def tables(n, cnt, hh):
        half = n // 2 - 1 if n % 2 == 0 else n // 2
        idx = np.arange(cnt)
        xi = (np.pi / n) * np.where(idx <= half, idx, idx - n)
        kpm = np.sin(xi) / hh
        return kpm, kpm * np.exp(1j * xi)
    kpm0, kp0 = tables(N, N, h[0])
    kpm1, kp1 = tables(ny, ny, h[1])
    kpm2, kp2 = tables(10, nzc, h[2])
    c10, c20, scale = -1.0 / 0.9, -1.0 / (0.9 * (1 + 0.9 / 1.1)), 1.0 / N
    y = x.copy()
    args = [np.ascontiguousarray(a) for a in (kpm0, kp0.view(np.float64), kpm1, kp1.view(np.float64), kpm2,
                                               kp2.view(np.float64))]
    assert emu.emu_xfused(N, P(y.view(np.float64)), ny, nzc, nzf, ctypes.c_double(scale), ctypes.c_double(c10),
                          ctypes.c_double(c20), *[P(a) for a in args]) == 0
    F = np.fft.fft(x, axis=1) * scale
    jj, kk = np.divmod(np.arange(ncols), nzc)
    K0, K1, K2 = kp0[:, None], kp1[jj][None, :], kp2[kk][None, :]
    n2 = (kpm0 ** 2)[:, None] + (kpm1[jj] ** 2)[None, :] + (kpm2[kk] ** 2)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        c1, c2 = c10 / n2, c20 / (n2 * n2)
        sdot = F[0] * K0 + F[1] * K1 + F[2] * K2
        G = np.stack([c1 * F[0] + c2 * sdot * (-np.conj(K0)), c1 * F[1] + c2 * sdot * (-np.conj(K1)),
                      c1 * F[2] + c2 * sdot * (-np.conj(K2))])
    G[:, 0, 0] = 0.0                      # zero frequency (kx = ky = kz = 0)
    live = (kk < nzf)[None, None, :]
    G = np.where(live, G, F)              # padding columns: scaled spectrum passed through
    ref = np.fft.ifft(G, axis=1) * N
    assert np.abs(y - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.parametrize("nx,ny", [(8, 16), (32, 8), (64, 256)])
def test_x_contiguous_layout_chain(emu, nx, ny):
    """Forward y pass writing [zc/8][y][x][8], the fused x pass on that layout, inverse y pass reading it back (the chain
    Solver::fft_g0_chain takes on large grids) against numpy on the plain layout: same numbers as the plain chain, only
    the addresses between the passes differ."""
    rng = np.random.default_rng(nx * ny)
    nzc, nzf = 16, 14
    x = rng.standard_normal((3, nx, ny, nzc)) + 1j * rng.standard_normal((3, nx, ny, nzc))
    xl = np.zeros((3, nzc // 8, ny, nx, 8), dtype=np.complex128)
    for c in range(3):
        assert emu.emu_strided_xlayout(ny, -1, P(np.ascontiguousarray(x[c]).view(np.float64)), P(xl[c].view(np.float64)), nx, nzc,
                                       ctypes.c_double(1.0)) == 0
    Fy = np.fft.fft(x, axis=2)
    as_plain = lambda a: a.transpose(0, 3, 2, 1, 4).reshape(3, nx, ny, nzc)     # [c][zt][y][x][8] -> [c][x][y][zc]
    assert np.abs(as_plain(xl) - Fy).max() / np.abs(Fy).max() < 1e-14
    h = [1.0 / (2 * nx), 2.0 / (2 * ny), 0.5 / (2 * 26)]

    def tables(n, cnt, hh):
        half = n // 2 - 1 if n % 2 == 0 else n // 2
        idx = np.arange(cnt)
        xi = (np.pi / n) * np.where(idx <= half, idx, idx - n)
        kpm = np.sin(xi) / hh
        return kpm, kpm * np.exp(1j * xi)
    kpm0, kp0 = tables(nx, nx, h[0])
    kpm1, kp1 = tables(ny, ny, h[1])
    kpm2, kp2 = tables(26, nzc, h[2])
    c10, c20, scale = -1.0 / 0.9, -1.0 / (0.9 * (1 + 0.9 / 1.1)), 1.0 / (nx * ny)
    args = [np.ascontiguousarray(a) for a in (kpm0, kp0.view(np.float64), kpm1, kp1.view(np.float64), kpm2, kp2.view(np.float64))]
    assert emu.emu_xfused_xlayout(nx, P(xl.view(np.float64)), ny, nzc, nzf, ctypes.c_double(scale), ctypes.c_double(c10),
                                  ctypes.c_double(c20), *[P(a) for a in args]) == 0
    F = np.fft.fft(Fy, axis=1) * scale
    K0, K1, K2 = kp0[:, None, None], kp1[None, :, None], kp2[None, None, :]
    n2 = (kpm0 ** 2)[:, None, None] + (kpm1 ** 2)[None, :, None] + (kpm2 ** 2)[None, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        c1, c2 = c10 / n2, c20 / (n2 * n2)
        sdot = F[0] * K0 + F[1] * K1 + F[2] * K2
        G = np.stack([c1 * F[0] + c2 * sdot * (-np.conj(K0)), c1 * F[1] + c2 * sdot * (-np.conj(K1)),
                      c1 * F[2] + c2 * sdot * (-np.conj(K2))])
    G[:, 0, 0, 0] = 0.0
    live = (np.arange(nzc) < nzf)[None, None, None, :]
    G = np.where(live, G, F)
    Gx = np.fft.ifft(G, axis=1) * nx
    assert np.abs(as_plain(xl) - Gx).max() / np.abs(Gx).max() < 1e-12
    back = np.zeros((3, nx, ny, nzc), dtype=np.complex128)
    for c in range(3):
        assert emu.emu_strided_xlayout(ny, +1, P(xl[c].view(np.float64)), P(back[c].view(np.float64)), nx, nzc, ctypes.c_double(1.0)) == 0
    ref = np.fft.ifft(Gx, axis=2) * ny
    assert np.abs(back - ref).max() / np.abs(ref).max() < 1e-12


@pytest.mark.parametrize("ny,nz", [(16, 16), (16, 32), (16, 64), (16, 128), (32, 16), (32, 32), (32, 64), (32, 128), (64, 16),
                                   (64, 32), (64, 64), (64, 128), (128, 16), (128, 32), (128, 64), (128, 128), (256, 16), (256, 32),
                                   (256, 64)])
def test_plane_kernels(emu, ny, nz):
    """fg_fft_plane.h: z and y transforms of whole planes in one kernel (the plane in LDS), forward and inverse, against numpy
    and -- bit for bit -- against the separate z and y passes."""
    rng = np.random.default_rng(ny * 1000 + nz)
    npl, nzc = 3, nz // 2 + 1
    x = rng.standard_normal((npl, ny, nz))
    buf = np.full((npl, ny, 2 * nzc), np.nan)
    buf[:, :, :nz] = x
    assert emu.emu_plane(ny, nz, -1, P(buf), npl) == 0
    ref = np.fft.fft(np.fft.rfft(x, axis=2), axis=1)
    assert np.abs(buf.view(np.complex128) - ref).max() / np.abs(ref).max() < 1e-14
    b2 = np.full((npl, ny, 2 * nzc), np.nan)
    b2[:, :, :nz] = x
    assert emu.emu_r2c(nz, P(b2), ctypes.c_long(npl * ny)) == 0
    assert emu.emu_strided(ny, -1, P(b2), nzc, npl, ctypes.c_double(1.0)) == 0
    assert np.array_equal(b2, buf)
    # inverse of an arbitrary (non-Hermitian) spectrum: FFTW c2r semantics
    X = rng.standard_normal((npl, ny, nzc)) + 1j * rng.standard_normal((npl, ny, nzc))
    b3 = X.copy().view(np.float64).copy()
    assert emu.emu_plane(ny, nz, +1, P(b3), npl) == 0
    ref3 = np.fft.irfft(np.fft.ifft(X, axis=1) * ny, n=nz, axis=2) * nz
    assert np.abs(b3[:, :, :nz] - ref3).max() / np.abs(ref3).max() < 1e-14
    b4 = X.copy().view(np.float64).copy()
    assert emu.emu_strided(ny, +1, P(b4), nzc, npl, ctypes.c_double(1.0)) == 0
    assert emu.emu_c2r(nz, P(b4), ctypes.c_long(npl * ny)) == 0
    assert np.array_equal(b4[:, :, :nz], b3[:, :, :nz])


# ---------------------------------------------------------------------------------------------------------------------
# Stockham tile kernels for lengths with factors 2, 3, 5, 7, 11, 13 (fg_fft_smooth.h): the decimal grid sizes
SMOOTH = [6, 10, 12, 15, 18, 20, 25, 30, 36, 49, 50, 60, 75, 77, 91, 100, 120, 121, 125, 143, 150, 169, 200, 240, 250, 300, 360,
          400, 480, 500, 600, 640, 700, 720, 900, 960, 1000, 1001, 1100, 1250]


def _plan(buf):
    return {"lines": int(buf[0]), "threads": int(buf[1]), "radices": [int(v) for v in buf[3:3 + buf[2]]]}


@pytest.mark.parametrize("N", SMOOTH)
@pytest.mark.parametrize("d", [-1, 1])
def test_smooth_strided_c2c(emu, N, d):
    rng = np.random.default_rng(N)
    ncols, nouter = 43, 2   # ragged last tile (8-, 16- and 32-column tiles)
    x = rng.standard_normal((nouter, N, ncols)) + 1j * rng.standard_normal((nouter, N, ncols))
    y = x.copy()
    plan = np.zeros(8, dtype=np.int32)
    assert emu.emu_smooth_strided(N, d, P(y.view(np.float64)), ncols, nouter, ctypes.c_double(0.5),
                                  plan.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == 0
    pl = _plan(plan)
    # the planner's contract: the radices multiply to N, every pass has at most one butterfly per thread, the image fits the LDS
    assert int(np.prod(pl["radices"])) == N and len(pl["radices"]) <= 4
    rounds = lambda r: 1 if r > 10 else min(8, 20 // r)   # smooth_rounds: butterflies of a small radix one thread may own
    assert all(N // r * pl["lines"] <= pl["threads"] * rounds(r) and r <= (32 if pl["threads"] == 256 else 16) for r in pl["radices"])
    assert N * pl["lines"] * 16 <= 156 * 1024
    ref = (np.fft.fft(x, axis=1) if d < 0 else np.fft.ifft(x, axis=1) * N) * 0.5
    assert np.abs(y - ref).max() / np.abs(ref).max() < 2e-14


@pytest.mark.parametrize("nz", [2 * n for n in SMOOTH])
def test_smooth_r2c_c2r(emu, nz):
    rng = np.random.default_rng(nz)
    nrows = 19   # ragged last tile
    nzc = nz // 2 + 1
    x = rng.standard_normal((nrows, nz))
    buf = np.full((nrows, 2 * nzc), np.nan)
    buf[:, :nz] = x
    plan = np.zeros(8, dtype=np.int32)
    assert emu.emu_smooth_z(nz, 1, P(buf), ctypes.c_long(nrows), plan.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == 0
    pl = _plan(plan)
    rounds = lambda r: 1 if r > 10 else min(8, 20 // r)
    assert int(np.prod(pl["radices"])) == nz // 2 and all(nz // 2 // r * pl["lines"] <= pl["threads"] * rounds(r) for r in pl["radices"])
    ref = np.fft.rfft(x, axis=1)
    assert np.abs(buf.view(np.complex128) - ref).max() / np.abs(ref).max() < 2e-14
    X = rng.standard_normal((nrows, nzc)) + 1j * rng.standard_normal((nrows, nzc))   # non-Hermitian: DC / Nyquist imaginary parts ignored
    buf = X.copy().view(np.float64).copy()
    assert emu.emu_smooth_z(nz, 0, P(buf), ctypes.c_long(nrows), None) == 0
    ref = np.fft.irfft(X, n=nz, axis=1) * nz
    assert np.abs(buf[:, :nz] - ref).max() / np.abs(ref).max() < 2e-14


@pytest.mark.parametrize("nz", [3, 5, 9, 15, 25, 27, 35, 45, 75, 77, 99, 105, 125, 143, 175, 225, 243, 375, 625])
def test_smooth_odd_rows(emu, nz):
    """odd nz: no packed-real trick -- the rows run through the same passes as nz complex points (before: O(nz^2) sums)"""
    rng = np.random.default_rng(nz)
    nrows = 19
    nzc = nz // 2 + 1
    x = rng.standard_normal((nrows, nz))
    buf = np.full((nrows, 2 * nzc), np.nan)
    buf[:, :nz] = x
    assert emu.emu_smooth_z(nz, 1, P(buf), ctypes.c_long(nrows), None) == 0
    ref = np.fft.rfft(x, axis=1)
    assert np.abs(buf.view(np.complex128) - ref).max() / np.abs(ref).max() < 2e-14
    X = rng.standard_normal((nrows, nzc)) + 1j * rng.standard_normal((nrows, nzc))   # the imaginary part of the DC bin is ignored
    buf = X.copy().view(np.float64).copy()
    assert emu.emu_smooth_z(nz, 0, P(buf), ctypes.c_long(nrows), None) == 0
    ref = np.fft.irfft(X, n=nz, axis=1) * nz
    assert np.abs(buf[:, :nz] - ref).max() / np.abs(ref).max() < 2e-14


def test_smooth_planner(emu):
    """the decimal sizes in two passes of large radices (one butterfly per thread, 8-column tiles, 256 threads); 1000 in three
    passes of 1024 threads; lengths with a prime factor above 13 have no plan"""
    want = {100: [10, 10], 200: [8, 5, 5], 300: [10, 10, 3], 400: [10, 10, 4], 500: [10, 10, 5], 120: [12, 10], 240: [16, 15],
            480: [10, 8, 6], 600: [25, 24], 1000: [10, 10, 10], 144: [12, 12], 96: [12, 8]}
    for N, radices in want.items():
        x = np.zeros((1, N, 1), dtype=np.complex128)
        plan = np.zeros(8, dtype=np.int32)
        assert emu.emu_smooth_strided(N, -1, P(x.view(np.float64)), 1, 1, ctypes.c_double(1.0),
                                      plan.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == 0
        pl = _plan(plan)
        assert pl["threads"] == (1024 if N == 1000 else 256) and int(np.prod(pl["radices"])) == N, (N, pl)
        assert N <= 256 or pl["radices"] == radices, (N, pl)   # (the wide tiles of short lines shift the radices: more butterflies per pass)
        assert pl["lines"] == (32 if N <= 128 else (16 if N <= 256 else 8)), (N, pl)   # image <= 64 KB for the wide tiles
    x = np.zeros((1, 34, 1), dtype=np.complex128)
    assert emu.emu_smooth_strided(34, -1, P(x.view(np.float64)), 1, 1, ctypes.c_double(1.0), None) == 1   # 34 = 2 * 17


@pytest.mark.parametrize("joint", [1, 0])
@pytest.mark.parametrize("N", [20, 36, 100, 120, 200, 225, 240, 300, 360, 400, 480, 500, 600])
def test_smooth_fused_x_pass(emu, N, joint):
    """The tile kernels' fused x pass (x transform, 1/N, Green operator G0OperatorFourierStaggeredGeneral F:19834-19927, inverse x
    transform) thread by thread on the host, in both forms: one image per component, and the three components on one joint
    image (every pass once over three times the butterflies, several butterflies of a small radix per thread)."""
    rng = np.random.default_rng(N + joint)
    ny, nzc, nzf = 3, 9, 7           # 27 columns: ragged last tile, two padding columns per row
    ncols = ny * nzc
    x = rng.standard_normal((3, N, ncols)) + 1j * rng.standard_normal((3, N, ncols))
    h = [1.0 / (2 * N), 2.0 / (2 * ny), 0.5 / (2 * 12)]

    def tables(n, cnt, hh):
        half = n // 2 - 1 if n % 2 == 0 else n // 2
        idx = np.arange(cnt)
        xi = (np.pi / n) * np.where(idx <= half, idx, idx - n)
        kpm = np.sin(xi) / hh
        return kpm, kpm * np.exp(1j * xi)
    kpm0, kp0 = tables(N, N, h[0])
    kpm1, kp1 = tables(ny, ny, h[1])
    kpm2, kp2 = tables(12, nzc, h[2])
    c10, c20, scale = -1.0 / 0.9, -1.0 / (0.9 * (1 + 0.9 / 1.1)), 1.0 / N
    y = x.copy()
    args = [np.ascontiguousarray(a) for a in (kpm0, kp0.view(np.float64), kpm1, kp1.view(np.float64), kpm2, kp2.view(np.float64))]
    plan = np.zeros(12, dtype=np.int32)
    rc = emu.emu_smooth_xfused(N, P(y.view(np.float64)), ny, nzc, nzf, ctypes.c_double(scale), ctypes.c_double(c10),
                               ctypes.c_double(c20), *[P(a) for a in args], joint,
                               plan.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if not joint and N > 416:
        assert rc == 1      # three 8-column images of such lines do not fit the LDS: fused on the joint 4-column image only
        return
    assert rc == 0
    lines, threads, npass, pj, cap = (int(v) for v in plan[:5])
    radices = [int(v) for v in plan[5:5 + npass]]
    assert int(np.prod(radices)) == N and threads in (256, 512, 1024)
    if not joint:
        assert pj == 0 and cap == 0 and threads != 512
    elif pj:   # the planner's joint form: three components' columns as the lines of one image
        assert pj == 3 and lines in (12, 24, 48) and threads in (256, 512) and cap in (20, 32)
        assert threads == 256 or (max(radices) <= 20 and cap == 20)
    F = np.fft.fft(x, axis=1) * scale
    jj, kk = np.divmod(np.arange(ncols), nzc)
    K0, K1, K2 = kp0[:, None], kp1[jj][None, :], kp2[kk][None, :]
    n2 = (kpm0 ** 2)[:, None] + (kpm1[jj] ** 2)[None, :] + (kpm2[kk] ** 2)[None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        c1, c2 = c10 / n2, c20 / (n2 * n2)
        sdot = F[0] * K0 + F[1] * K1 + F[2] * K2
        G = np.stack([c1 * F[0] + c2 * sdot * (-np.conj(K0)), c1 * F[1] + c2 * sdot * (-np.conj(K1)),
                      c1 * F[2] + c2 * sdot * (-np.conj(K2))])
    G[:, 0, 0] = 0.0                      # zero frequency
    ref = np.fft.ifft(G, axis=1) * N
    live = kk < nzf                       # (padding columns of a row carry no data)
    assert np.abs(y[:, :, live] - ref[:, :, live]).max() / np.abs(ref[:, :, live]).max() < 1e-12


def test_smooth_fused_x_plans(emu):
    """which lengths run the fused x pass on the joint image (no more passes than one image per component), and with what"""
    got = {}
    for N in (100, 120, 200, 240, 300, 360, 400, 480, 500, 600, 800):
        x = np.zeros((3, N, 8), dtype=np.complex128)
        z = np.zeros(max(N, 8))
        zc = np.zeros(2 * max(N, 8))
        plan = np.zeros(12, dtype=np.int32)
        assert emu.emu_smooth_xfused(N, P(x.view(np.float64)), 1, 8, 8, ctypes.c_double(1.0), ctypes.c_double(1.0),
                                     ctypes.c_double(1.0), P(z), P(zc), P(z), P(zc), P(z), P(zc), 1,
                                     plan.ctypes.data_as(ctypes.POINTER(ctypes.c_int))) == 0
        got[N] = (int(plan[0]), int(plan[3]), int(plan[4]), int(plan[1]), [int(v) for v in plan[5:5 + plan[2]]])
    print(got)
    assert got[200] == (24, 3, 20, 256, [20, 10])   # (lines, joint components, values per thread, threads, radices)
    assert got[100] == (48, 3, 20, 256, [10, 10])
    assert got[300] == (24, 3, 20, 512, [20, 15])   # 7 200 points: 512 threads with <= 20 values each
    assert got[400] == (24, 3, 20, 512, [20, 20])
    assert got[500] == (12, 3, 20, 512, [10, 10, 5])   # 4-column tiles from 420 points on


def test_plan_kernel_tables_match_the_planner(emu):
    """every entry of the plan kernels' tables (fg_fft_smooth_plans.h: kernels built for one plan each) is the plan the
    planner makes for that length -- an entry that is not would silently fall back to the class kernels"""
    n = ctypes.c_int(0)
    assert emu.emu_plan_table_mismatches(ctypes.byref(n)) == 0
    assert n.value == 76 + 80 + 52 + 80
