"""pmx -- MI355X-native vectorised Pac-Man Capture-the-Flag environment step.

The directory is named after the reference repository (`pacman-marl-2025_amd`); import it as `pmx` (alias package at
the repo root)."""
from . import _lib
from ._lib import PmxError
from .layout import Layout, get_layout
from . import maze_generator
from .vec_env import PmxVecEnv, legal_list, make_state

__all__ = ["PmxError", "Layout", "get_layout", "PmxVecEnv", "legal_list", "make_state"]
