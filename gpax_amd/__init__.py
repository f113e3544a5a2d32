"""
gpax_amd — MI355X-native exact-GP hot path behind the gpax API surface
(ExactGP / viGP fit()/predict(), gpax.kernels, gpax.utils).  Python host code over libgpx
(hand-written HIP for gfx950) through ctypes.  No JAX, no PyTorch, no CPU fallback.
"""
from . import acquisition, kernels, priors, utils
from .infer import dist
from .infer.primitives import deterministic, plate, sample, seed
from .models import ExactGP, MeasuredNoiseGP, VarNoiseGP, vExactGP, viGP, viSparseGP

__version__ = "0.1.0"
__all__ = ["ExactGP", "vExactGP", "viGP", "viSparseGP", "MeasuredNoiseGP", "VarNoiseGP", "kernels", "priors", "utils", "acquisition", "dist",
           "sample", "plate", "deterministic", "seed"]
