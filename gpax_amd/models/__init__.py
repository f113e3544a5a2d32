from .gp import ExactGP
from .linreg import LinReg
from .mngp import MeasuredNoiseGP
from .sparse_gp import viSparseGP
from .vgp import vExactGP
from .vigp import viGP

__all__ = ["ExactGP", "vExactGP", "viGP", "viSparseGP", "MeasuredNoiseGP", "LinReg"]
