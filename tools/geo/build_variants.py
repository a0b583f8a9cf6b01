"""Build experiment variants of the geometry kernel: libenvidr_amd.so re-linked with geometry_pass.hip compiled under
different -D switches (tools/geo/variants/<name>.so; selected at run time through ENVIDR_AMD_LIB)."""
import subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from envidr_amd import build as B

VARIANTS = {
    "k32": [],
    "k32_a1": ["-DENVIDR_GEO_AHEAD=1"],
    "k32_a3": ["-DENVIDR_GEO_AHEAD=3"],
    "k64": ["-DENVIDR_GEO_KERNEL32=0", "-DENVIDR_GEO_KERNEL16=0"],
    "k32only": ["-DENVIDR_GEO_KERNEL16=0"],
    "k16": [],
    "k16_a1": ["-DENVIDR_GEO_AHEAD=1"],
    "k16_a3": ["-DENVIDR_GEO_AHEAD=3"],
    "k16_w16_a1": ["-DENVIDR_GEO_E16_WAVES=16", "-DENVIDR_GEO_AHEAD=1"],
    "k16_w8": ["-DENVIDR_GEO_E16_WAVES=8"],
    "rays_dbg1": ["-DENVIDR_GEO_RAYS_DEBUG=1"],
    "rays_dbg2": ["-DENVIDR_GEO_RAYS_DEBUG=2"],
}

SPLIT_VARIANTS = {
    "split": [],
    "split_nostream": ["-DENVIDR_SPLIT_DEBUG=1"],
    "split_nobarrier": ["-DENVIDR_SPLIT_DEBUG=2"],
    "split_noconv": ["-DENVIDR_SPLIT_DEBUG=3"],
    "split_loadsonly": ["-DENVIDR_SPLIT_DEBUG=4"],
    "split_writesonly": ["-DENVIDR_SPLIT_DEBUG=5"],
    "split_g4": ["-DENVIDR_SPLIT_GROUP=4"],
    "split_pin0": ["-DENVIDR_SPLIT_PIN=0"],
    "split_pin1": ["-DENVIDR_SPLIT_PIN=1"],
    "split_pin2": ["-DENVIDR_SPLIT_PIN=2"],
    "split_unpiped": ["-DENVIDR_SPLIT_PIPED=0"],
    "split_pf4": ["-DENVIDR_SPLIT_AHEAD=4"],
}

def main(names):
    B.build(verbose=False)
    out = ROOT / "tools" / "geo" / "variants"
    out.mkdir(exist_ok=True)
    for name in names or VARIANTS:
        src = "shade_split" if name in SPLIT_VARIANTS else "geometry_pass"
        objs = [str(o) for o in sorted((B.CSRC / "build").glob("*.o")) if o.stem != src]
        flags = SPLIT_VARIANTS[name] if name in SPLIT_VARIANTS else VARIANTS[name]
        obj = out / f"{name}.o"
        cmd = [B.hipcc(), *B.HIPCC_FLAGS, *flags, "-Rpass-analysis=kernel-resource-usage", "-c", str(B.CSRC / f"{src}.hip"), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(name, "FAILED\n", r.stderr[-3000:]); continue
        info = [l.split("remark:")[1].strip().replace("[-Rpass-analysis=kernel-resource-usage]", "") for l in r.stderr.splitlines()
                if any(k in l for k in ("VGPRs:", "VGPRs Spill", "ScratchSize", "LDS Size"))]
        lib = out / f"{name}.so"
        r = subprocess.run([B.hipcc(), f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", *objs, str(obj), "-o", str(lib)],
                           capture_output=True, text=True)
        print(name, "|", " ".join(info), "|", "link ok" if r.returncode == 0 else r.stderr[-500:])

if __name__ == "__main__":
    main(sys.argv[1:])
