"""pmx -- MI355X-native vectorised Pac-Man Capture-the-Flag environment step.

The directory is named after the reference repository (`pacman-marl-2025_amd`); import it as `pmx` (alias package at
the repo root)."""
from . import _lib
from ._lib import PmxError
from .layout import Layout, get_layout
from . import maze_generator
from .vec_env import PmxVecEnv, legal_list, make_state



def __getattr__(name):
    # the drop-in class is imported lazily: it pulls in torch-side helpers that the C-ABI-only users do not need
    if name == "gymPacMan_parallel_env":
        from .gym_env import gymPacMan_parallel_env
        return gymPacMan_parallel_env
    raise AttributeError(name)


__all__ = ["PmxError", "Layout", "get_layout", "PmxVecEnv", "legal_list", "make_state", "gymPacMan_parallel_env"]
