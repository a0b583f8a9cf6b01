"""Static properties of the built device code that the measured numbers rest on, checked without a GPU (tools/kernel_resources.py reads the
kernel descriptors' metadata out of libenvidr_amd.so): the frame's kernels use no scratch memory and spill no vector register, the
split-precision kernel of two waves per SIMD stays within 256 registers without AGPRs, and one workgroup's LDS fits the 160 KiB of a CU."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))


def _rows():
    import kernel_resources as kr
    rows = []
    for _, blob in kr.code_objects(ROOT / "envidr_amd" / "libenvidr_amd.so"):
        rows += kr.kernels_of(blob)
    names = kr.demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["pretty"] = n
    return rows


def test_the_frames_kernels_do_not_spill_and_fit_the_cu():
    rows = _rows()
    assert len(rows) > 400
    I = lambda r, k: int(r.get(k, 0) or 0)
    by = lambda key: [r for r in rows if key in r["pretty"]]
    for key in ("k_geo_eval32<", "k_shade_samples<5, 8, 0, false, false>", "k_shade_samples<4, 5, 0, false, false>", "k_env_split2<", "k_geo_rays<",
                "k_composite_records", "k_place_records", "k_table_scatter_lds<3, 2,", "k_linear_rows<", "k_linear_weight_grad<false, true>"):
        sel = by(key)
        assert sel, key
        for r in sel:
            assert I(r, "private_segment_fixed_size") == 0 and I(r, "vgpr_spill_count") == 0, (r["pretty"][:120], r)
    for r in by("k_env_split2<"):
        # eight waves of a workgroup, two per SIMD: 256 registers each, and above 256 the compiler would move accumulators to AGPRs
        assert I(r, "vgpr_count") <= 256 and I(r, "agpr_count") == 0 and I(r, "max_flat_workgroup_size") == 512, r
    assert max(I(r, "group_segment_fixed_size") for r in rows) <= 160 * 1024
    assert max(I(r, "vgpr_count") for r in rows) <= 512
