"""MI355X-native MC-GRA adjacency-optimisation hot path.

The directory name follows the task's layout (``mc-gra_amd``); it is loaded as
the package ``mc_gra_amd`` by ``mcgra_loader.load()`` at the repo root.
"""
from . import _lib  # noqa: F401  (raises if libmcgra_hip.so is missing: no CPU fallback)
from .base_attack import BaseAttack  # noqa: F401
from .engine import AttackEngine  # noqa: F401
from .topology_attack import PGDAttack  # noqa: F401

__all__ = ["BaseAttack", "PGDAttack", "AttackEngine"]
