"""Structured GP: a probabilistic mean function and custom kernel priors written the way the reference writes them
(gpax README "Structured GP" / tests/test_gp.py:25-38: functions that call `numpyro.sample`) — here with
`gpax_amd.sample` — followed by a predictive sweep sharded over every GPU of the node from this one process
(`device="all"`: RCCL broadcast of the inputs, RCCL gather of the results; the vmap axis of gp.py:392-395).

    python examples/custom_priors_node_sweep.py
"""
import numpy as np

import gpax_amd as gpax
from gpax_amd import dist


def piecewise(x, params):  # a mean function with a kink at t (x has shape (N, 1))
    x = x[:, 0]
    return np.where(x < params["t"], params["a"] * x, params["a"] * params["t"] + params["b"] * (x - params["t"]))


def piecewise_priors():
    t = gpax.sample("t", dist.Uniform(0.5, 2.5))
    a = gpax.sample("a", dist.Normal(0.0, 2.0))
    b = gpax.sample("b", dist.Normal(0.0, 2.0))
    return {"t": t, "a": a, "b": b}


def kernel_priors():
    length = gpax.sample("k_length", dist.Gamma(2.0, 4.0))
    scale = gpax.sample("k_scale", dist.HalfNormal(0.5))
    return {"k_length": length, "k_scale": scale}


def main(num_warmup=200, num_samples=200, verbose=True):
    rng = np.random.default_rng(1)
    X = np.sort(rng.uniform(0.0, 3.0, 80))
    truth = lambda x: np.where(x < 1.7, 1.2 * x, 1.2 * 1.7 - 0.8 * (x - 1.7)) + 0.15 * np.sin(9 * x)
    y = truth(X) + 0.08 * rng.standard_normal(X.size)
    X_test = np.linspace(0.0, 3.0, 150)

    k1, k2 = gpax.utils.get_keys(1)
    m = gpax.ExactGP(1, "Matern", mean_fn=piecewise, mean_fn_prior=piecewise_priors, kernel_prior=kernel_priors,
                     noise_prior_dist=dist.HalfNormal(0.2))
    m.fit(k1, X, y, num_warmup=num_warmup, num_samples=num_samples, progress_bar=verbose, print_summary=verbose)
    y_one, s_one = m.predict(k2, X_test, n=2)                 # this GPU
    y_all, s_all = m.predict(k2, X_test, n=2, device="all")   # every GPU of the node, one process
    t_hat = float(np.median(m.get_samples()["t"]))
    rmse = float(np.sqrt(np.mean((y_all - truth(X_test)) ** 2)))
    if verbose:
        print(f"kink at {t_hat:.2f} (truth 1.70); RMSE {rmse:.3f}; node sweep == single-GPU sweep: "
              f"{np.array_equal(s_one, s_all)}")
    return dict(rmse=rmse, t=t_hat, same=bool(np.array_equal(s_one, s_all) and np.array_equal(y_one, y_all)))


if __name__ == "__main__":
    main()
