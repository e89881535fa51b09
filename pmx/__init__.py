"""Import alias: the product package lives in the directory `pacman-marl-2025_amd/`, whose name is not a valid
Python identifier.  `import pmx` executes that package's __init__ under the name `pmx`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "pacman-marl-2025_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
