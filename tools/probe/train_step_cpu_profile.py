"""Where the HOST's time of a training step goes (the step is host-bound: tools/train_step_bench.py): cProfile over 10 steps with the backward
pass run on the calling thread (torch.autograd.set_multithreading_enabled(False)) so that it is seen."""
import sys, cProfile, pstats, io
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.argv = [sys.argv[0], "3"]
import runpy, torch
torch.autograd.set_multithreading_enabled(False)
ns = runpy.run_path(str(Path(__file__).resolve().parents[1] / "train_step_bench.py"))
step = ns["step"]
for _ in range(3): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats(key).print_stats(45); print(buf.getvalue()[:9000])
