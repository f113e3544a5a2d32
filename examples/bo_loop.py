"""Bayesian optimisation loop — steps A - D of the reference README ("Active learning and Bayesian optimization"):
fit a GP on the measured points, compute the UCB acquisition on the unmeasured grid, measure at its argmax, repeat;
here to localise the minimum of a 1-D "black box" with as few measurements as possible, on the MI355X path.

    python examples/bo_loop.py
"""
import numpy as np

import gpax_amd as gpax


def black_box(x):
    return np.sin(3.0 * x) + 0.6 * np.cos(7.0 * x) + 0.25 * x ** 2  # global minimum near x = -0.47 on [-2, 2]


def main(num_seed_points=4, num_steps=10, num_warmup=100, num_samples=100, beta=4.0, verbose=True):
    rng = np.random.default_rng(1)
    X_grid = np.linspace(-2.0, 2.0, 201)
    idx_measured = list(rng.choice(X_grid.size, num_seed_points, replace=False))
    noise_sd = 0.02
    measure = lambda i: float(black_box(X_grid[i]) + noise_sd * rng.standard_normal())
    y_measured = [measure(i) for i in idx_measured]
    history = []
    for step in range(num_steps):
        rng_key, rng_key_predict = gpax.utils.get_keys(step)
        unmeasured = np.setdiff1d(np.arange(X_grid.size), idx_measured)
        X_measured, X_unmeasured = X_grid[idx_measured], X_grid[unmeasured]
        gp_model = gpax.ExactGP(1, kernel='Matern')
        gp_model.fit(rng_key, X_measured, np.asarray(y_measured), num_warmup=num_warmup, num_samples=num_samples,
                     progress_bar=False, print_summary=False)                                          # A
        acq = gpax.acquisition.UCB(rng_key_predict, gp_model, X_unmeasured, beta=beta, maximize=False,
                                   noiseless=True)                                                     # B
        next_point_idx = int(unmeasured[int(np.argmax(acq))])                                          # C, D
        idx_measured.append(next_point_idx)
        y_measured.append(measure(next_point_idx))
        best = int(np.argmin(y_measured))
        history.append((X_grid[next_point_idx], X_grid[idx_measured[best]], y_measured[best]))
        if verbose:
            print(f"step {step + 1:2d}: measured x = {X_grid[next_point_idx]:+.3f}; best so far f({X_grid[idx_measured[best]]:+.3f})"
                  f" = {y_measured[best]:+.4f}")
    x_true = X_grid[np.argmin(black_box(X_grid))]
    if verbose:
        print(f"true minimum on the grid: f({x_true:+.3f}) = {black_box(x_true):+.4f} after {len(idx_measured)} measurements")
    return dict(history=history, x_best=history[-1][1], y_best=history[-1][2], x_true=float(x_true),
                y_true=float(black_box(x_true)), n_measured=len(idx_measured))


if __name__ == "__main__":
    main()
