"""envidr_amd -- MI355X-native (gfx950) implementation of ENVIDR's render hot path.

Sub-packages mirror the reference's operator surface (same names, arguments and error behaviour):
`raymarching`, `hashencoder`, `gridencoder`, `freqencoder`, `shencoder`, `ide_encoder`, `encoding`,
and `nerf` (renderer / network / render loop).  All of them call hand-written HIP kernels through
the C ABI declared in include/envidr_amd.h; there is no CPU fallback.
"""
__version__ = "0.1.0"
