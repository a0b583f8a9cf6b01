"""Where the time of an env-sphere frame (run_sph, fused form) goes: per-stage HIP-event times at 400 x 400 and 800 x 800 (run through gpurun)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from envidr_amd import scenes  # noqa: E402
from envidr_amd.nerf.network import NeRFNetwork  # noqa: E402
from envidr_amd.nerf.options import EnvOptions, neural_renderer_options  # noqa: E402
from envidr_amd.nerf.render_func import sph_ray  # noqa: E402


def main():
    dev = torch.device("cuda")
    opt = neural_renderer_options(env_sph_radius=0.95 * 0.8)
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="hashgrid", encoding_dir=opt.encoding_dir, bound=opt.bound, cuda_ray=False, density_scale=1, min_near=opt.min_near,
                    density_thresh=opt.density_thresh, bg_radius=opt.bg_radius, use_sdf=True, hidden_dim=opt.hidden_dim, num_layers=opt.num_layers,
                    num_layers_color=opt.num_layers_color, hidden_dim_color=opt.hidden_dim_color, num_levels=opt.num_levels,
                    geo_feat_dim=opt.geo_feat_dim, opt=opt, env_opt=EnvOptions()).to(dev).eval()
    with torch.no_grad():
        m.encoder.embeddings.uniform_(-0.1, 0.1)
        m.sdf_density.beta.fill_(0.005)
    mat = {"roughness": 0.3, "metallic": 0.2, "color": [20 / 255, 70 / 255, 160 / 255, 1.0]}
    for res in (400, 800):
        ro, rd = (torch.from_numpy(a).to(dev) for a in scenes.camera_rays(res, res, theta=123.0, phi=10.0, radius=4.0, scale=0.8))
        N = res * res
        fr = m.fused_sph_renderer(3, mat)
        bg = torch.ones(N, 3, device=dev)

        def stages():
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(7)]
            ev[0].record()
            nears, fars, mask = fr.sphere_intersections(ro, rd, opt.env_sph_radius)
            ev[1].record()
            hit = torch.nonzero(mask).squeeze(-1).to(torch.int32)
            M = int(hit.shape[0])
            slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
            slot[hit.long()] = torch.arange(M, dtype=torch.int32, device=dev)
            zoff = torch.linspace(-0.011, 0.011, 12, device=dev)
            ev[2].record()
            xyz, dirs, z = fr.shell_samples(ro, rd, hit, nears.reshape(-1).contiguous(), zoff, 0.002, None)
            ev[3].record()
            geo = fr.geometry_eval(xyz, want=("sigma", "normal", "geo_feat", "roughness"))
            ev[4].record()
            sh = fr.shade(geo["normal"], dirs, geo["geo_feat"], geo["roughness"], None)
            ev[5].record()
            fr.composite_shell(geo["sigma"], z, sh["c_diffuse"], sh["c_specular"], geo["normal"], geo["roughness"], slot, nears.reshape(-1).contiguous(),
                               fars.max().reshape(1), bg, 0.002, want=("diffuse_image", "specular_image"))
            ev[6].record()
            torch.cuda.synchronize()
            return M, [ev[i].elapsed_time(ev[i + 1]) for i in range(6)]
        for _ in range(3):
            M, t = stages()
        names = ["envidr_sphere_intersections", "compaction + slots (torch, one host sync)", "envidr_shell_samples", "envidr_geometry_eval", "envidr_shade_samples",
                 "envidr_composite_shell (+ fars.max)"]
        print(f"{res}x{res}: {M} hit rays, {12 * M} samples; " + "; ".join(f"{n} {x:.3f} ms" for n, x in zip(names, t)) + f"; sum {sum(t):.3f} ms")


if __name__ == "__main__":
    with torch.no_grad():
        main()
