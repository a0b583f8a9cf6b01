"""The sanitizer driver (tools/asan_driver.py) and its torch stand-in (tools/asan_shim) rehearsed without a GPU: with ENVIDR_ASAN_SHIM_DRY=1
the stand-in's "device" blocks are host memory and every launch is refused by the runtime ("no ROCm-capable device"), so what runs is exactly
the host code on the way to each launch -- operand marshalling of all operator cases, the weight packers, the frame buffers -- on the stand-in's
tensors.  The driver gets one GPU call per round; this keeps it from meeting its own bugs there."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_sanitizer_driver_rehearsal_reaches_every_launch():
    env = dict(os.environ, ENVIDR_ASAN_SHIM_DRY="1", ENVIDR_AMD_LIB=os.environ.get("ENVIDR_AMD_LIB", str(ROOT / "envidr_amd" / "libenvidr_amd.so")))         # (tools/host_sanitize.sh points it at the host-sanitized build)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "asan_driver.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "ASAN DRIVER DONE: no failure" in r.stdout, r.stdout[-3000:]
    assert " FAILED" not in r.stdout and "not carried by the torch stand-in" not in r.stdout, r.stdout[-3000:]
    line = next(l for l in r.stdout.splitlines() if l.startswith("REHEARSAL without a GPU"))
    assert int(line.split(":")[1].split()[0]) >= 400, line           # every operator case + the sections behind them got to a launch
    cases_line = next(l for l in r.stdout.splitlines() if l.startswith("operator cases through the sanitizer build"))
    assert int(cases_line.split(":")[1].split()[0]) >= 378 and ", 0 failed" in cases_line, cases_line
    for section in ("dense layer 32 -> 64", "geometry-only frame re-shaded",
                    "persistent single-kernel frame", "shade[f16x2_v1]"):
        assert section in r.stdout, section
