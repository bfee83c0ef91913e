"""Importable alias of the `gym-guidance-collision-avoidance-single_amd/` package
(its directory name is not a valid Python identifier)."""
import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "gym-guidance-collision-avoidance-single_amd")
__path__ = [_impl]
with open(_os.path.join(_impl, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_impl, "__init__.py"), "exec"))
del _os, _f, _impl
