"""hash_encode_forward on 7.7 M random points (the standalone gather DESIGN.md 3.5 prices): ours or (`ref`) the reference's own kernel compiled
by hipcc for this GPU -- a few calls and nothing else, for rocprofv3 --pmc / --kernel-trace (tools/gpu_gather_pmc.sh).  `frame`: a frame's
own samples (ray-major) instead of random points."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import _lib, scenes

dev = torch.device("cuda:0")
sc = scenes.toaster_scene()
table = torch.from_numpy(sc.table).to(dev)
offsets = torch.from_numpy(np.ascontiguousarray(sc.offsets, np.int32)).to(dev)
S = float(np.log2(sc.per_level_scale))
M = 7_700_000
torch.manual_seed(0)
x01 = torch.rand(M, 3, device=dev)
if "frame" in sys.argv:
    # consecutive samples of consecutive rays: points along segments, neighbours 1/128 of the box apart
    rays = M // 12
    o = torch.rand(rays, 1, 3, device=dev) * 0.6 + 0.2
    d = torch.nn.functional.normalize(torch.randn(rays, 1, 3, device=dev), dim=-1)
    t = torch.arange(12, device=dev).view(1, 12, 1) / 128.0
    x01 = (o + d * t).clamp(0, 1).reshape(-1, 3)[:M].contiguous()
out = torch.empty(16, M, 2, device=dev)
args = (x01, table, offsets, out, M, 3, 2, 16, S, 16, 0, None)
if "ref" in sys.argv:
    from oracle import clib
    lib = clib.ref_hip()
    call = lambda: lib.call("hash_encode_forward", *[a.data_ptr() if isinstance(a, torch.Tensor) else a for a in args])
else:
    call = lambda: _lib.call("hash_encode_forward", *args)
for _ in range(4):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(4):
    call()
e1.record(); torch.cuda.synchronize()
print(f"{'reference kernel' if 'ref' in sys.argv else 'envidr_amd'} hash_encode_forward, {M} {'frame-ordered' if 'frame' in sys.argv else 'random'} points: {e0.elapsed_time(e1) / 4:.3f} ms")
