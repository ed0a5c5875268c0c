"""Importable name for the package kept in ``fdtd-2d_amd/`` (the directory name the
project layout asks for is not a Python identifier).  Submodules resolve there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "fdtd-2d_amd")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f, _real, _os
