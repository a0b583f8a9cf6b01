"""`ide_encoder` as the reference's code imports it (ide_encoder/__init__.py:1): IntegratedDirEncoder on the HIP operator."""
from envidr_amd.ide_encoder import IntegratedDirEncoder      # noqa: F401
