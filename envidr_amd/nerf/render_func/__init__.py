from .cuda_ray import run_cuda  # noqa: F401
