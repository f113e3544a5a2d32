"""Reconstruction of a sparsely sampled image with a variational sparse GP — the workflow of the reference's
examples/gpax_viGP.ipynb (preprocess_sparse_image -> viSparseGP.fit -> predict_in_batches), on the MI355X path.

    python examples/sparse_image.py
"""
import numpy as np

import gpax_amd as gpax


def main(size=96, keep=0.15, num_steps=200, verbose=True):
    rng = np.random.default_rng(1)
    ii, jj = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    image = 1.5 + np.sin(ii / 9.0) * np.cos(jj / 13.0) + 0.3 * np.exp(-((ii - 60) ** 2 + (jj - 30) ** 2) / 150.0)
    sparse = np.where(rng.uniform(size=image.shape) < keep, image, 0.0)  # zeros = missing pixels

    X_train, y_train, X_full = gpax.utils.preprocess_sparse_image(sparse)
    rng_key, rng_key_predict = gpax.utils.get_keys()
    model = gpax.viSparseGP(2, kernel='Matern', lengthscale_prior_dist=gpax.priors.gamma_dist(5, 0.5))
    model.fit(rng_key, X_train, y_train - y_train.mean(), num_steps=num_steps, step_size=0.05, inducing_points_ratio=0.1,
              progress_bar=verbose, print_summary=verbose)
    mean, var = model.predict_in_batches(rng_key_predict, X_full, batch_size=1000)
    recon = (mean + y_train.mean()).reshape(image.shape)
    rmse = float(np.sqrt(np.mean((recon - image) ** 2)))
    if verbose:
        print(f"{X_train.shape[0]} of {image.size} pixels observed, {model.Xu.shape[0]} inducing points: "
              f"reconstruction RMSE {rmse:.4f} (image sd {image.std():.3f})")
    return dict(rmse=rmse, image_sd=float(image.std()), recon=recon, var=var.reshape(image.shape))


if __name__ == "__main__":
    main()
