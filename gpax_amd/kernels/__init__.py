from .kernels import MaternKernel, RBFKernel, add_jitter, get_kernel, kernel_name

__all__ = ["RBFKernel", "MaternKernel", "get_kernel", "add_jitter", "kernel_name"]
