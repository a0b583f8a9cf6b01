#!/bin/bash
# AddressSanitizer pass over the library on the GPU (through gpurun), then the synchronisation stress loop on the product library.
#   tools/gpu_asan.sh <tag>        needs tools/asan/libenvidr_amd_asan.so (python tools/build_asan.py, in the container)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cp tools/asan/build.log $OUT/asan_build.log 2>/dev/null
# 1. the stress loop on the product library (no sanitizer: full speed, 1000 repetitions)
timeout 1500 python tools/stress_sync.py 1000 > $OUT/stress_sync.txt 2>&1; echo "stress rc=$?" >> $OUT/stress_sync.txt; tail -8 $OUT/stress_sync.txt
# 2. the sanitizer build: device code instrumented (xnack+), host runtime preloaded
if [ -f tools/asan/libenvidr_amd_asan.so ]; then
  RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  export HSA_XNACK=1 ENVIDR_AMD_LIB=$GRAFT_REPO_ROOT/tools/asan/libenvidr_amd_asan.so LD_PRELOAD=$RT
  export ASAN_OPTIONS=detect_leaks=0:halt_on_error=0:protect_shadow_gap=0:log_path=$OUT/asan_report
  timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_scatter_gpu.py tests/test_split_gpu.py -q -x -k "not refhip" -p no:cacheprovider > $OUT/asan_pytest.txt 2>&1; echo "pytest rc=$?" >> $OUT/asan_pytest.txt
  tail -15 $OUT/asan_pytest.txt
  # one 48x48 frame of each frame path
  timeout 900 python - > $OUT/asan_frames.txt 2>&1 <<PY
import torch
from envidr_amd import scenes
from envidr_amd.fused import FusedOptions, FusedRenderer
ro, rd = (torch.from_numpy(a).cuda() for a in scenes.camera_rays(48, 48))
r = FusedRenderer.from_scene(scenes.toaster_scene())
for name, fn in (("geometry pipeline + record shading (fp32)", lambda: r.render_frame(ro, rd, 0.2, out={})),
                 ("split-precision shading", lambda: r.render_frame(ro, rd, 0.2, out={}, env_precision="f16x2")),
                 ("persistent single-kernel frame", lambda: r.render(ro, rd, 0.2, extras=True, out={})),
                 ("geometry cache", lambda: r.render_cached(r.cache_geometry(ro, rd), 0.3, out={}))):
    img = fn()["image"]
    torch.cuda.synchronize()
    print(name, "ok", float(img.mean()))
p = FusedRenderer.from_scene(scenes.lego_scene(), FusedOptions(dir_sh_degree=4))
print("no-environment family ok", float(p.render_frame(ro, rd, None, out={})["image"].mean()))
PY
  echo "frames rc=$?" >> $OUT/asan_frames.txt; tail -8 $OUT/asan_frames.txt
  ls $OUT/asan_report* 2>/dev/null | head; for f in $OUT/asan_report*; do [ -f "$f" ] && head -40 "$f"; done
  echo "sanitizer reports: $(ls $OUT/asan_report* 2>/dev/null | wc -l)" | tee $OUT/asan_summary.txt
else
  echo "tools/asan/libenvidr_amd_asan.so not built" | tee $OUT/asan_summary.txt
fi
