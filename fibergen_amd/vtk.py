"""Legacy-VTK result files in the reference's layout.

Restates VTKCubeWriter (src/fibergen.cpp:5714-6071) and LSSolver::writeVTK / writeVTKPhase
(src/fibergen.cpp:23250-23451): DATASET STRUCTURED_POINTS with DIMENSIONS (nx+1, ny+1, nz+1), CELL_DATA of
nx*ny*nz cells, one SCALARS block per phase fraction (phi_<name>) and per strain / stress component
(epsilon_11 ... sigma_12), the displacement as one VECTORS block "u".  Cell order is x fastest, then y, then z
(writeXYSlice, F:6041-6069).  Binary files hold big-endian values of the result type (float unless
<restype>double</restype>, F:26552), a newline in front of every field after the first and one at the end
(F:5910-5912, F:5790-5793).  ASCII files print one cell per line with the default ostream precision.
"""
import numpy as np

_VOIGT_NAMES = ("11", "22", "33", "23", "13", "12")


class VTKCubeWriter:
    def __init__(self, filename, shape, dims, x0, binary=True, dtype="float"):
        if dtype not in ("float", "double"):
            raise RuntimeError("data type not supported")
        self._fs = open(filename, "wb")
        self._shape = tuple(int(n) for n in shape)
        self._dims = dims
        self._x0 = x0
        self._binary = bool(binary)
        self._dtype = dtype
        self._nfields = 0

    @staticmethod
    def _num(v):
        return "%g" % v  # operator<< of a double with the default precision of 6

    def write_mesh(self):
        nx, ny, nz = self._shape
        sp = [d / n for d, n in zip(self._dims, self._shape)]
        head = "# vtk DataFile Version 2.0\nfibergen\n" + ("BINARY\n" if self._binary else "ASCII\n")
        head += "DATASET STRUCTURED_POINTS\n"
        head += "DIMENSIONS %d %d %d\n" % (nx + 1, ny + 1, nz + 1)
        head += "ORIGIN %s %s %s\n" % tuple(self._num(v) for v in self._x0)
        head += "SPACING %s %s %s\n" % tuple(self._num(v) for v in sp)
        head += "CELL_DATA %d\n" % (nx * ny * nz)
        self._fs.write(head.encode("ascii"))

    def _begin(self, name, vectors):
        head = "\n" if (self._binary and self._nfields > 0) else ""
        head += ("VECTORS" if vectors else "SCALARS") + " " + name + " " + self._dtype + "\n"
        if not vectors:
            head += "LOOKUP_TABLE default\n"
        self._fs.write(head.encode("ascii"))
        self._nfields += 1

    def _payload(self, comps):
        # comps: list of [nx][ny][nz] arrays -> cell-major (z slowest, x fastest), components interleaved
        a = np.stack([np.asarray(c, dtype=np.float64).transpose(2, 1, 0) for c in comps], axis=-1)
        if self._binary:
            self._fs.write(a.astype(">f4" if self._dtype == "float" else ">f8").tobytes())
        else:
            a = a.astype(np.float32 if self._dtype == "float" else np.float64).reshape(-1, len(comps))
            lines = "\n".join(" ".join("%g" % v for v in row) for row in a)
            self._fs.write((lines + "\n").encode("ascii"))

    def write_scalar(self, name, field):
        self._begin(name, False)
        self._payload([field])

    def write_vector(self, name, field3):
        self._begin(name, True)
        self._payload(list(field3))

    def close(self):
        if self._binary:
            self._fs.write(b"\n")
        self._fs.close()


def write_results(filename, shape, dims, x0, phase_names, phi, epsilon, sigma, u, binary=True, dtype="float",
                  mode="elasticity"):
    """LSSolver::writeVTK  F:23317-23451 (elasticity and heat / porous branches)."""
    w = VTKCubeWriter(filename, shape, dims, x0, binary, dtype)
    w.write_mesh()
    for m, name in enumerate(phase_names):
        w.write_scalar("phi_" + name, phi[m])
    ncomp = 6 if mode == "elasticity" else 3
    for c in range(ncomp):
        w.write_scalar("epsilon_" + _VOIGT_NAMES[c], epsilon[c])
    for c in range(ncomp):
        w.write_scalar("sigma_" + _VOIGT_NAMES[c], sigma[c])
    if mode == "elasticity":
        w.write_vector("u", u)
    else:
        w.write_scalar("T" if mode == "heat" else "p", u[0])
    w.close()


def write_phase(filename, shape, dims, x0, name, phi, binary=True, dtype="float"):
    """LSSolver::writeVTKPhase  F:23296-23314"""
    w = VTKCubeWriter(filename, shape, dims, x0, binary, dtype)
    w.write_mesh()
    w.write_scalar("phi_" + name, phi)
    w.close()


def read_legacy(filename):
    """Minimal reader for the files above (tests and round trips): returns (header dict, {name: array})
    with arrays of shape [ncomp][nx][ny][nz]."""
    raw = open(filename, "rb").read()
    pos = 0

    def line():
        nonlocal pos
        e = raw.index(b"\n", pos)
        s = raw[pos:e].decode("ascii")
        pos = e + 1
        return s
    head = {"version": line(), "title": line(), "format": line(), "dataset": line()}
    dims = [int(v) - 1 for v in line().split()[1:]]
    head["shape"] = tuple(dims)
    head["origin"] = [float(v) for v in line().split()[1:]]
    head["spacing"] = [float(v) for v in line().split()[1:]]
    ncell = int(line().split()[1])
    assert ncell == dims[0] * dims[1] * dims[2]
    binary = head["format"] == "BINARY"
    fields = {}
    while pos < len(raw):
        if raw[pos:pos + 1] == b"\n":
            pos += 1
            continue
        kind, name, dtype = line().split()
        nc = 3 if kind == "VECTORS" else 1
        if kind == "SCALARS":
            assert line() == "LOOKUP_TABLE default"
        if binary:
            dt = np.dtype(">f4" if dtype == "float" else ">f8")
            nbytes = ncell * nc * dt.itemsize
            a = np.frombuffer(raw[pos:pos + nbytes], dtype=dt).astype(np.float64)
            pos += nbytes
        else:
            vals = []
            for _ in range(ncell):
                vals.extend(float(v) for v in line().split())
            a = np.array(vals)
        a = a.reshape(dims[2], dims[1], dims[0], nc).transpose(3, 2, 1, 0)
        fields[name] = np.ascontiguousarray(a)
    return head, fields
