from ..priors.priors import (auto_lognormal_priors, auto_normal_priors, gamma_dist, halfnormal_dist, lognormal_dist,
                             normal_dist, uniform_dist)
from . import threefry
from .utils import (enable_x64, get_keys, initialize_inducing_points, preprocess_sparse_image, random_sample_dict,
                    rng_from_key, set_fn, split_dict, split_in_batches, split_key)

# the reference re-exports the prior helpers from gpax.utils as well (gpax/utils/__init__.py)
__all__ = ["enable_x64", "get_keys", "initialize_inducing_points", "preprocess_sparse_image", "random_sample_dict",
           "rng_from_key", "set_fn", "split_dict", "split_in_batches", "split_key", "normal_dist", "lognormal_dist",
           "halfnormal_dist", "gamma_dist", "uniform_dist", "auto_normal_priors", "auto_lognormal_priors", "threefry"]
