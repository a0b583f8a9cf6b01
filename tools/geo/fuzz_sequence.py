"""Host-side state machine of FusedRenderer.render_frame under a random call sequence: batch sizes coming and going (buffers keyed by
N, at most four kept), tiny capacity guesses (FrameOverflow redo), frames left in flight (wait=False) and collected later, tags,
masks, geometry-only passes, the trimming of over-sized buffers -- every result must equal the one a fresh renderer gives.
Run on the GPU box:  python tools/geo/fuzz_sequence.py [calls] [seed]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np, torch
from envidr_amd import scenes
from envidr_amd.fused import FrameOverflow, FusedRenderer

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")
scene = scenes.toaster_scene()
r = FusedRenderer.from_scene(scene, device=dev)
fresh = FusedRenderer.from_scene(scene, device=dev)
views = {}
def view(side, theta):
    key = (side, theta)
    if key not in views:
        ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(side, side, theta=theta, phi=-25.0))
        views[key] = (ro, rd)
    return views[key]
KEYS = ("image", "depth", "weights_sum", "normal_image")
bad = 0
inflight = []          # (description, result dict (live buffers), expected clones)
def collect():
    global bad
    try:
        r.check_frames()
    except FrameOverflow:
        inflight.clear()      # frames in flight when one overflowed are void by contract; nothing to compare
        return
    torch.cuda.synchronize()
    for desc, res, want in inflight:
        for k in want:
            if not torch.equal(res[k], want[k]):
                bad += 1; print("in-flight frame differs:", desc, k)
    inflight.clear()
for c in range(calls):
    side = int(rng.choice([5, 16, 33, 64, 90, 128]))
    theta = float(rng.choice([0.0, 40.0, 200.0]))
    ro, rd = view(side, theta)
    N = ro.shape[0]
    rot = float(rng.choice([0.0, 1.0, 2.5]))
    geo = rng.random() < 0.15
    mask = (torch.arange(N, device=dev) % 3 != 0) if rng.random() < 0.2 else None
    tag = str(rng.choice(["", "a", "b"]))
    hint = float(rng.choice([0.05, 1.0, 20.0, 20.0]))
    wait = rng.random() < 0.6
    fresh.__dict__.pop("_frames", None)
    want = fresh.render_frame(ro, rd, rot, geometry_only=geo, ray_mask=mask, image_width=side)
    want = {k: want[k].clone() for k in KEYS if k in want}
    desc = f"call {c}: side {side} theta {theta} rot {rot} geo {geo} mask {mask is not None} tag {tag!r} hint {hint} wait {wait}"
    try:
        if inflight and rng.random() < 0.5:
            collect()
        out = {}
        res = r.render_frame(ro, rd, rot, out=out, geometry_only=geo, ray_mask=mask, tag=tag, samples_per_ray_hint=hint, wait=wait,
                             image_width=side if rng.random() < 0.7 else 0)
        if wait:
            torch.cuda.synchronize()
            for k in want:
                if not torch.equal(res[k], want[k]):
                    bad += 1; print("differs:", desc, k)
            inflight.clear()          # wait=True checked every frame in flight
        else:
            inflight.append((desc, res, want))
            if len(inflight) > 3:
                collect()
    except FrameOverflow:
        inflight.clear()
    except Exception as e:      # noqa: BLE001
        bad += 1; print("EXCEPTION", desc, type(e).__name__, str(e)[:200]); inflight.clear()
collect()
print(f"{calls} calls, {bad} findings; buffer sets kept: {sorted(r._frames)}")
sys.exit(1 if bad else 0)
