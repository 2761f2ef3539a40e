"""fibergen_amd -- MI355X-native Lippmann-Schwinger FFT homogenisation (fibergen's hot path).

`FG` mirrors the reference's Python class (fibergen.FG, src/fibergen.cpp:27142-27187);
`LSSolver` is the object view of the C ABI in include/fibergen_amd.h.
"""
from .solver import LSSolver  # noqa: F401

__all__ = ["LSSolver", "FG"]


def __getattr__(name):
    if name == "FG":
        from .fg import FG
        return FG
    raise AttributeError(name)
