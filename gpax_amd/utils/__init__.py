from .utils import (enable_x64, get_keys, initialize_inducing_points, preprocess_sparse_image, random_sample_dict,
                    rng_from_key, split_dict, split_in_batches, split_key)

__all__ = ["enable_x64", "get_keys", "initialize_inducing_points", "preprocess_sparse_image", "random_sample_dict",
           "rng_from_key", "split_dict", "split_in_batches", "split_key"]
