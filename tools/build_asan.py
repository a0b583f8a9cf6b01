"""TEST TOOLING: an AddressSanitizer build of the whole library for gfx950 (device code instrumented: xnack+, -fsanitize=address) ->
tools/asan/libenvidr_amd_asan.so, selected at run time through ENVIDR_AMD_LIB (tools/gpu_asan.sh runs the GPU tests under it with HSA_XNACK=1).
Every translation unit is compiled on its own (in parallel); one that the sanitizer cannot build is reported and left out of the link --
the entry points it defines are then missing from the library, and the tests that need them are deselected by gpu_asan.sh.

    python tools/build_asan.py [-j N]
"""
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from envidr_amd import build as B  # noqa: E402

OUT = ROOT / "tools" / "asan"
FLAGS = ["--offload-arch=gfx950:xnack+", "-fsanitize=address", "-shared-libsan", "-gline-tables-only", "-O3", "-std=c++20", "-fPIC",
         "-ffp-contract=off", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wno-unused-function", "-Wno-unknown-pragmas"]


def compile_one(src: Path):
    obj = OUT / (src.stem + ".o")
    r = subprocess.run([B.hipcc(), *FLAGS, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    return src.name, obj, r.returncode, r.stderr[-1500:]


def main():
    jobs = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else 8
    OUT.mkdir(exist_ok=True)
    with ThreadPoolExecutor(jobs) as ex:
        results = list(ex.map(compile_one, B.sources()))
    objs, log = [], []
    for name, obj, rc, err in results:
        log.append(f"{name}: {'ok' if rc == 0 else 'FAILED'}" + ("" if rc == 0 else "\n    " + err.replace("\n", "\n    ")))
        if rc == 0:
            objs.append(str(obj))
    lib = OUT / "libenvidr_amd_asan.so"
    r = subprocess.run([B.hipcc(), "--offload-arch=gfx950:xnack+", "-fsanitize=address", "-shared-libsan", "-shared", "-fPIC", "-fno-gpu-rdc", *objs, "-o", str(lib)],
                       capture_output=True, text=True)
    log.append(f"link: {'ok' if r.returncode == 0 else 'FAILED ' + r.stderr[-1500:]}")
    (OUT / "build.log").write_text("\n".join(log) + "\n")
    print("\n".join(log))


if __name__ == "__main__":
    main()
