from .kernels import (MaternKernel, PeriodicKernel, RBFKernel, add_jitter, get_kernel, kernel_name,
                      square_scaled_distance)

__all__ = ["RBFKernel", "MaternKernel", "PeriodicKernel", "get_kernel", "add_jitter", "kernel_name",
           "square_scaled_distance"]
