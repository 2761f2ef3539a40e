"""Drop-in module name: `import fibergen; fg = fibergen.FG()` resolves to the MI355X path."""
from fibergen_amd.fg import FG  # noqa: F401
from fibergen_amd.solver import LSSolver  # noqa: F401
