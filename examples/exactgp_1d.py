"""Fully Bayesian GP regression on noisy 1-D data — the workflow of the reference README
(gpax README.md "Gaussian Process" section: get_keys -> ExactGP.fit -> predict), on the MI355X path.

    python examples/exactgp_1d.py
"""
import numpy as np

import gpax_amd as gpax


def main(num_warmup=300, num_samples=300, verbose=True):
    rng = np.random.default_rng(0)
    X = np.sort(rng.uniform(-1.0, 1.0, 60))
    f = lambda x: np.sin(8 * x) * np.exp(-x ** 2)
    y = f(X) + 0.1 * rng.standard_normal(X.size)
    X_test = np.linspace(-1.2, 1.2, 200)

    rng_key, rng_key_predict = gpax.utils.get_keys()
    gp_model = gpax.ExactGP(1, kernel='Matern', noise_prior_dist=gpax.priors.halfnormal_dist(0.5))
    gp_model.fit(rng_key, X, y, num_warmup=num_warmup, num_samples=num_samples, progress_bar=verbose,
                 print_summary=verbose)
    y_pred, y_sampled = gp_model.predict(rng_key_predict, X_test, n=10)

    # acquisition for the next measurement (gpax.acquisition.EI, README "Bayesian optimization")
    acq = gpax.acquisition.EI(rng_key_predict, gp_model, X_test, maximize=True, noiseless=True)
    inside = np.abs(X_test) <= 1.0
    rmse = float(np.sqrt(np.mean((y_pred[inside] - f(X_test[inside])) ** 2)))
    spread = y_sampled.reshape(-1, X_test.size).std(0)
    if verbose:
        print(f"RMSE inside the data range: {rmse:.3f};  predictive sd inside / outside: "
              f"{spread[inside].mean():.3f} / {spread[~inside].mean():.3f};  next point (EI): {X_test[np.argmax(acq)]:.3f}")
    return dict(rmse=rmse, sd_in=float(spread[inside].mean()), sd_out=float(spread[~inside].mean()),
                y_pred=y_pred, y_sampled=y_sampled, acq=acq)


if __name__ == "__main__":
    main()
