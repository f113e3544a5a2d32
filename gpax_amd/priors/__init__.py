from .priors import (auto_lognormal_priors, auto_normal_priors, auto_priors, gamma_dist, halfnormal_dist, lognormal_dist,
                     normal_dist, place_gamma_prior, place_halfnormal_prior, place_lognormal_prior, place_normal_prior,
                     place_uniform_prior, uniform_dist)

__all__ = ["normal_dist", "lognormal_dist", "halfnormal_dist", "gamma_dist", "uniform_dist", "auto_priors",
           "auto_normal_priors", "auto_lognormal_priors", "place_normal_prior", "place_lognormal_prior",
           "place_halfnormal_prior", "place_uniform_prior", "place_gamma_prior"]
