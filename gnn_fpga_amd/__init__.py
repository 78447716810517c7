"""Import alias for the package directory `gnn-fpga_amd/` (a hyphen is not a valid
Python identifier, so `import gnn_fpga_amd` resolves here and runs that package)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "gnn-fpga_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
