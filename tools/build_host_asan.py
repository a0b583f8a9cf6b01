"""Host-side sanitizer build of the C-ABI library: every translation unit with -fsanitize=address,undefined on the HOST half only (the
device half is the normal gfx950 code; the device-side ASAN build is tools/build_asan.py).  It needs no GPU: the CPU test suite drives the
packers, the descriptor validation and the error paths of the library through ctypes, and tools/host_sanitize.sh runs that suite against
this build with the sanitizer runtime preloaded.  Output: tools/asan_host/libenvidr_amd_hostasan.so (git- and gpurun-ignored)."""
import subprocess, sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from envidr_amd import build as B
OUT = Path(__file__).resolve().parent / 'asan_host'
OUT.mkdir(exist_ok=True)
FLAGS = ["--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-shared-libsan", "-g1", "-O1", "-std=c++20", "-fPIC", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wno-unused-function", "-Wno-unknown-pragmas"]
def one(src):
    obj = OUT / (src.stem + ".o")
    r = subprocess.run([B.hipcc(), *FLAGS, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
    return src.name, obj, r.returncode, r.stderr[-800:]
with ThreadPoolExecutor(8) as ex:
    res = list(ex.map(one, B.sources()))
for n, o, rc, err in res:
    print(n, "ok" if rc == 0 else "FAILED\n" + err)
objs = [str(o) for _, o, rc, _ in res if rc == 0]
r = subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-fsanitize=address,undefined", "-shared-libsan", "-shared", "-fPIC", "-fno-gpu-rdc", *objs, "-o", str(OUT / "libenvidr_amd_hostasan.so")], capture_output=True, text=True)
print("link", r.returncode, r.stderr[-800:])
