from .sphere_harmonics import SHEncoder, sh_encode  # noqa: F401
