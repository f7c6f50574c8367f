"""Import alias: the package directory is ``ga-ddpg_amd/`` (not a valid Python identifier), so
``import ga_ddpg_amd`` resolves here and redirects the package search path to it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ga-ddpg_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
