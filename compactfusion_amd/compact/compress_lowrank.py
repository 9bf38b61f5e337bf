"""Import-path mirror of `xfuser/compact/compress_lowrank.py`: `subspace_iter`, `svd` (implemented in lowrank.py)."""
from .lowrank import subspace_iter, svd  # noqa: F401
