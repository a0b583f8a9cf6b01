import sys; sys.path.insert(0, "/root/repo")
import torch, time
import torch.nn as nn
from envidr_amd import fused
torch.manual_seed(0)
dev = torch.device("cuda")
for k, h in fused.ENV_MLP_SHAPES:
    net = nn.ModuleList([nn.Linear(k, h), nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, 12)]).to(dev)
    for M in (1, 63, 64, 65, 4097, 300001):
        x = torch.randn(M, k, device=dev)
        with torch.no_grad():
            y = fused.env_mlp_forward(net, x)
            r = x
            for i, l in enumerate(net):
                r = l(r)
                if i != 3: r = torch.relu(r)
        err = ((y - r).norm() / r.norm()).item()
        print(k, h, M, "rel-L2 %.2e" % err, "max abs %.2e" % (y - r).abs().max().item())
        assert err < 2e-6
k, h = 72, 256
net = nn.ModuleList([nn.Linear(k, h), nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, 12)]).to(dev)
M = 7_713_316
x = torch.randn(M, k, device=dev)
with torch.no_grad():
    for fn, name in ((lambda: fused.env_mlp_forward(net, x), "operator"),):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(name, "%.2f ms" % (dt * 1e3), "%.1f TFLOP/s" % (M * 2 * (72 * 256 + 2 * 256 * 256 + 256 * 12) / dt / 1e12))
    def torch_chain():
        r = x
        for i, l in enumerate(net):
            r = l(r)
            if i != 3: r = torch.relu(r)
        return r
    torch_chain(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): torch_chain()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("torch", "%.2f ms" % (dt * 1e3))
