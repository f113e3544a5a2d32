from .gp import ExactGP
from .hskgp import VarNoiseGP
from .linreg import LinReg
from .mngp import MeasuredNoiseGP
from .sparse_gp import viSparseGP
from .vgp import vExactGP
from .vigp import viGP

__all__ = ["ExactGP", "vExactGP", "viGP", "viSparseGP", "MeasuredNoiseGP", "VarNoiseGP", "LinReg"]
