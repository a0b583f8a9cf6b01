"""The reference-shaped import surface (envidr_amd.compat) without a GPU: names, argument counts, both installation ways."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_backends_have_the_pybind_names_and_arities():
    from envidr_amd import _lib
    from envidr_amd.compat.backends import EXTENSIONS, make_backend
    # reference */src/bindings.cpp: 11 + 3 + 2 + 2 + 2 functions
    assert {k: len(v) for k, v in EXTENSIONS.items()} == {"raymarching": 11, "hashencoder": 3, "gridencoder": 2, "freqencoder": 2, "shencoder": 2}
    for pkg, names in EXTENSIONS.items():
        m = make_backend(pkg)
        for n in names:
            fn = getattr(m, n)
            with pytest.raises(TypeError):
                fn()                                               # wrong arity, like pybind
            assert n in _lib.SIGNATURES
    # argument counts of the reference's headers (raymarching.h:7-18, hashencoder.h:13-15, gridencoder.h:12-13, ...)
    want = {"near_far_from_aabb": 7, "sph_from_ray": 5, "morton3D": 3, "morton3D_invert": 3, "packbits": 4, "get_scatter_idx": 3,
            "march_rays_train": 19, "composite_rays_train_forward": 13, "composite_rays_train_backward": 17, "march_rays": 18,
            "composite_rays": 13, "hash_encode_forward": 12, "hash_encode_backward": 14, "hash_encode_second_backward": 15,
            "grid_encode_forward": 13, "grid_encode_backward": 15, "freq_encode_forward": 6, "freq_encode_backward": 7,
            "sh_encode_forward": 6, "sh_encode_backward": 7}
    assert {n: len(_lib.SIGNATURES[n]) for n in want} == want


def test_both_installation_ways_resolve_the_reference_imports():
    code = r'''
import sys, os
sys.path.insert(0, %r)
import envidr_amd.compat as compat
# way 1: the reference's wrappers keep importing `<pkg>._ext._<pkg>`
compat.install_backends()
from raymarching._ext import _raymarching as b1
from hashencoder._ext import _hashencoder as b2
from gridencoder._ext import _gridencoder as b3
from freqencoder._ext import _freqencoder as b4
from shencoder._ext import _shencoder as b5
assert callable(b1.march_rays) and callable(b2.hash_encode_second_backward) and callable(b3.grid_encode_backward)
assert callable(b4.freq_encode_forward) and callable(b5.sh_encode_backward)
for k in [k for k in sys.modules if k.split(".")[0] in ("raymarching", "hashencoder", "gridencoder", "freqencoder", "shencoder")]:
    del sys.modules[k]
# way 2: the compat directory on sys.path shadows the reference's packages
sys.path.insert(0, os.path.dirname(compat.__file__))
import raymarching, hashencoder, gridencoder, freqencoder, shencoder, ide_encoder
assert "compat" in raymarching.__file__
for name in ("near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "get_scatter_idx", "march_rays_train",
             "composite_rays_train", "march_rays", "composite_rays"):            # raymarching.py:49,80,104,126,155,164,246,310,367,394
    assert callable(getattr(raymarching, name)), name
assert hashencoder.HashEncoder and gridencoder.GridEncoder and freqencoder.FreqEncoder and shencoder.SHEncoder and ide_encoder.IntegratedDirEncoder
from raymarching._ext import _raymarching
assert callable(_raymarching.composite_rays)
print("ok")
''' % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-3000:]
