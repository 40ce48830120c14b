"""Import alias: the package directory is ``distributed-matvec_amd/`` (not a valid Python
identifier), so ``import distributed_matvec_amd`` resolves its submodules from there."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "..", "distributed-matvec_amd"))

from .api import *  # noqa: F401,F403,E402
from . import api as _api  # noqa: E402

__all__ = _api.__all__
