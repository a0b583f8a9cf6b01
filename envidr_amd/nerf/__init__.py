"""Renderer / network / render loop mirroring the reference's `nerf` package on the HIP operators."""
