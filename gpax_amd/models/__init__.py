from .gp import ExactGP
from .vigp import viGP

__all__ = ["ExactGP", "viGP"]
