"""The HOST LOGIC of the reference-shaped operator path, checked without a GPU -- NOT a parity test of the kernels.

`NeRFNetwork.render()` with `fused=False` is Python on top of the C-ABI operators: the march / compact / composite loop of run_cuda, the
three passes of indirect rendering, the background sphere, the env-sphere mode's operator chain, the training branch with its autograd
wrappers (first and second backward of the hash encoder, the compositor's backward), the occupancy-grid maintenance, the checkpoint reader.
On a GPU box the `-m gpu` tests run all of it on the HIP operators against fixtures the imported reference produced.  Here the SAME test
bodies (imported from the GPU test modules, same fixtures, same tolerances) run on CPU tensors with `envidr_amd._lib.call` -- the one
function through which this layer reaches the library -- pointed at the oracle's C restatement of each operator (which tests/test_golden_cpu.py
and tests/test_oracle_pinning.py pin against the reference's own kernel bodies).  What a pass says: the Python layer of this tree drives the
operators the way the reference's does (argument order, buffer sizes, loop control, gradient plumbing).  What it does not say: anything
about the HIP kernels -- that is `-m gpu`.  The product itself has no CPU path: the fused renderers refuse a CPU device, and `_lib.call`
refuses CPU tensors (tests/test_abi_cpu.py); the substitution below exists only inside this file's fixture."""
import numpy as np
import pytest
import torch

from envidr_amd import scenes


@pytest.fixture
def oracle_operators(monkeypatch):
    from envidr_amd import _lib
    from envidr_amd.raymarching import raymarching as rm
    from oracle import clib

    def call(name, *args, stream=None):
        work = []
        for a in args:
            if isinstance(a, torch.Tensor):
                assert not a.is_cuda and a.is_contiguous(), name
                work.append(a.detach().numpy())          # (shares the tensor's memory: outputs land in place, as on the device)
            else:
                work.append(a)
        clib.oracle().call(name, *work)

    monkeypatch.setattr(_lib, "call", call)
    monkeypatch.setattr(rm, "_gpu", lambda t: t)
    # the GPU test bodies move their inputs with .cuda() and wait with torch.cuda.synchronize(): both are the identity here
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)


def test_the_product_refuses_cpu_tensors_without_the_substitution():
    from envidr_amd import _lib
    o, d, near, far = torch.zeros(4, 3), torch.ones(4, 3), torch.zeros(4), torch.zeros(4)
    with pytest.raises(_lib.EnvidrError, match="must live on the GPU"):
        _lib.call("near_far_from_aabb", o, d, torch.tensor([-1.0, -1, -1, 1, 1, 1]), 4, 0.2, near, far)


@pytest.mark.parametrize("tag", ["toaster_48", "toaster_rot_40"])
def test_operator_loop_renders_the_references_frames(oracle_operators, tag):
    import tests.test_dropin_gpu as D
    D.test_render_matches_reference_frames(D.build_model(scenes.toaster_scene()), tag, False)


def test_per_sample_chain(oracle_operators):
    import tests.test_dropin_gpu as D
    D.test_per_sample_chain_matches_reference(D.build_model(scenes.toaster_scene()))


@pytest.mark.parametrize("body", ["test_instance_norm_feature_activations_match_reference", "test_neus_geometric_init_and_skip_layer_variants_match_reference"])
def test_other_network_forms(oracle_operators, body):
    import tests.test_dropin_gpu as D
    getattr(D, body)()


@pytest.mark.parametrize("body", ["test_background_sphere_branch_matches_reference", "test_indirect_three_pass_matches_reference",
                                  "test_chunked_indirect_render_matches_reference", "test_indirect_with_an_object_box_matches_reference",
                                  "test_relight_with_shipped_checkpoints", "test_config1_no_env_network"])
def test_render_variants_through_the_operator_loop(oracle_operators, body):
    """background sphere; the three passes of indirect rendering (whole, chunked, with an object box); relighting through the checkpoint
    reader with the reference's shipped MLPs; BASELINE configs[1] (no environment network, SH-encoded directions)"""
    import tests.test_dropin_gpu as D
    getattr(D, body)(False)


@pytest.mark.parametrize("tag", ["toaster", "lego"])
def test_training_branch_forward_and_gradients(oracle_operators, tag):
    """march_rays_train -> autograd normals (create_graph) -> colours -> composite_rays_train, then the loss backward: every network's
    gradient, beta's, and the hash table's rows against the reference's own autograd (tests/golden/train_*.npz)"""
    import tests.test_train_gpu as T
    T.test_training_branch_forward_and_gradients_match_the_reference(tag)


def test_occupancy_grid_maintenance(oracle_operators):
    import tests.test_grid_gpu as G
    G.test_grid_maintenance_matches_reference()


@pytest.mark.parametrize("tag,normal", [("40", True), ("200", False)])
def test_env_sphere_operator_chain(oracle_operators, tag, normal):
    import tests.test_sph_gpu as S
    from tests import sph_case
    g = sph_case.load()
    S.test_env_sphere_render_matches_reference(g, sph_case.build_model(g, device="cpu"), False, tag, normal)


def test_env_sphere_operator_chain_staged(oracle_operators):
    import tests.test_sph_gpu as S
    from tests import sph_case
    g = sph_case.load()
    S.test_staged_env_sphere_render_matches_reference(g, sph_case.build_model(g, device="cpu"), False)


@pytest.mark.parametrize("tag", ["uniform", "resampled"])
def test_torch_only_render_function(oracle_operators, tag):
    """cuda_ray = False (tests/test_zz_plain_gpu.py, which has not met a GPU since its re-sampled fixture was re-generated): the same body,
    fixture and bounds with the oracle's encoders underneath"""
    import tests.test_zz_plain_gpu as P
    P.test_torch_only_render_function_matches_the_reference(P.build_plain_model(), tag)


def test_torch_only_render_function_staged_and_its_sampler(oracle_operators):
    import tests.test_zz_plain_gpu as P
    P.test_inverse_cdf_sampling_against_numpy()
    P.test_staged_chunks_render_the_same_frame(P.build_plain_model())


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_operator_loop_on_random_options_equals_the_oracles_frame_loop(oracle_operators, seed):
    """The fixtures pin the loop at the shipped options.  Here the loop's own knobs are drawn at random -- step-size growth, step limit,
    early-termination threshold, near plane, roughness / intensity scales, environment rotation, scene shape and weights -- and the frame
    is compared with oracle/py/render_oracle.py's restatement of the reference's loop (pinned to the reference's frames and integer
    schedules by tests/test_golden_cpu.py) on the same operators: what differs is only who drives them."""
    import tests.test_dropin_gpu as D
    from oracle.py import render_oracle as ro
    rng = np.random.default_rng(100 + seed)
    dt_gamma = float(rng.choice([0.0, 1 / 256, 1 / 128, 1 / 64]))
    max_steps = int(rng.choice([48, 128, 512, 1024]))
    T_thresh = float(rng.choice([1e-4, 1e-3, 2e-2]))
    min_near = float(rng.choice([0.05, 0.2, 0.5]))
    rough_scale, intensity, light = float(rng.uniform(0.5, 1.2)), float(rng.uniform(0.6, 1.3)), float(rng.uniform(0.7, 1.4))
    env_rot = None if seed % 2 else float(rng.uniform(-3, 3))
    shape = scenes.torus() if seed % 3 == 0 else None
    scene = scenes.toaster_scene(shape=shape, seed=20 + seed, sdf_bias=float(rng.choice([0.005, 0.03])), beta=float(rng.choice([0.01, 0.03])))
    model, opt = D.build_model(scene, dt_gamma=dt_gamma, max_steps=max_steps, T_thresh=T_thresh, min_near=min_near, roughness_scale=rough_scale,
                               intensity_scale=intensity, light_intensity_scale=light)
    H = W = 24
    rays_o, rays_d = scenes.camera_rays(H, W, theta=float(rng.uniform(0, 360)), phi=float(rng.uniform(-60, -10)))
    res = model.render(torch.from_numpy(rays_o)[None], torch.from_numpy(rays_d)[None], staged=True, bg_color=1, perturb=False, get_normal_image=True,
                       env_rot_radian=env_rot, fused=False, max_steps=max_steps, T_thresh=T_thresh, dt_gamma=dt_gamma)
    want = ro.render_rays(scene, rays_o, rays_d, ro.RenderOptions(ide_mode="exact", dt_gamma=dt_gamma, max_steps=max_steps, T_thresh=T_thresh, min_near=min_near,
                                                                   roughness_scale=rough_scale, intensity_scale=intensity, light_intensity_scale=light), env_rot)
    assert want["n_samples"] > 500
    from tests.util import rel_l2
    for key in D.KEYS:
        got = res[key].detach().numpy().reshape(H * W, -1)
        err = rel_l2(got, want[key].reshape(H * W, -1))
        assert err <= 2e-5, f"seed {seed} {key}: rel-L2 {err:.3e}"
    assert np.array_equal(res["depth"].numpy().reshape(-1), want["depth"])          # the integer side: same samples, same order of additions
