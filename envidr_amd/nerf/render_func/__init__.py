from .cuda_ray import fused_eligible, run_cuda  # noqa: F401
from .sph_ray import get_sphere_intersections, render_surface, run_sph  # noqa: F401
from .non_cuda_ray import run  # noqa: F401
