"""Encoder factory with the reference's string -> encoder mapping (encoding.py:46-91)."""
from __future__ import annotations


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, **kwargs):
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency":
        if multires == 0:
            return (lambda x, **kw: x), input_dim
        from .freqencoder import FreqEncoder
        encoder = FreqEncoder(input_dim=input_dim, degree=multires)
    elif encoding == "sphere_harmonics":
        from .shencoder import SHEncoder
        encoder = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding == "integrated_dir":
        from .ide_encoder import IntegratedDirEncoder
        encoder = IntegratedDirEncoder(input_dim=input_dim, deg_view=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        from .gridencoder import GridEncoder
        encoder = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                              log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                              gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners)
    elif encoding == "hashgrid_diff":
        from .hashencoder import HashEncoder
        encoder = HashEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, per_level_scale=2,
                              base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size,
                              desired_resolution=desired_resolution)
    else:
        raise NotImplementedError("Unknown encoding mode, choose from [None, frequency, sphere_harmonics, integrated_dir, "
                                  "hashgrid, hashgrid_diff, tiledgrid]")
    return encoder, encoder.output_dim
