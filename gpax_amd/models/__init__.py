from .gp import ExactGP
from .sparse_gp import viSparseGP
from .vgp import vExactGP
from .vigp import viGP

__all__ = ["ExactGP", "vExactGP", "viGP", "viSparseGP"]
