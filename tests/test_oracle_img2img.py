"""CPU: the oracle pieces behind the image-to-image tests (test infrastructure checking itself).

* oracle/txt2img.run(start_step=k) from the latents the full loop holds after step k-1 reproduces the rest of that loop exactly
  (the scheduler state of every kind restarts cleanly inside the schedule);
* modelgen's vae_encoder.onnx binds, name by name, to the torch restatement that is pinned on the reference's own export."""
import numpy as np
import pytest
import torch

from oracle import fixture_models as FM, onnx_io, schedulers as S, txt2img as T


@pytest.mark.parametrize("kind", ["euler", "euler_a", "ddim", "ddpm", "dpmpp"])
def test_loop_restarts_inside_the_schedule(kind):
    rng = np.random.default_rng(3)
    B, h, w, steps, k = 2, 6, 5, 9, 4
    ehs = rng.standard_normal((2 * B, 7, 8)).astype(np.float32)
    init = rng.standard_normal((B, 4, h, w)).astype(np.float32)
    noise = rng.standard_normal((steps + 3, B, 4, h, w)).astype(np.float32)
    wgt = rng.standard_normal((4, 4)).astype(np.float32) * 0.3

    def unet(x, t, e):   # any deterministic function of its inputs will do
        return (np.einsum("oc,nchw->nohw", wgt, x) + 0.01 * np.float32(t) / 1000 + e.mean(axis=(1, 2))[:, None, None, None] * 0.1).astype(np.float32)

    full = T.run(unet, S.make(kind), ehs, init, steps, 5.0, None, noise)
    nt = len(full["timesteps"])
    assert k < nt
    part = T.run(unet, S.make(kind), ehs, full["latents"][k - 1], steps, 5.0, None, noise, start_step=k)
    assert len(part["timesteps"]) == nt - k
    if kind == "dpmpp":   # reference quirk: always first order, so there is no history to lose
        assert np.array_equal(part["final"], full["final"])
    np.testing.assert_allclose(part["noise_pred"][0], full["noise_pred"][k], rtol=0, atol=0)
    np.testing.assert_allclose(part["final"], full["final"], rtol=1e-6, atol=1e-6)


def test_lms_restart_is_first_order_again():
    """LMS keeps derivative history: a run started inside the schedule begins with an empty history (diffusers' img2img does the same)."""
    rng = np.random.default_rng(4)
    ehs = rng.standard_normal((2, 3, 4)).astype(np.float32)
    init = rng.standard_normal((1, 4, 4, 4)).astype(np.float32)
    unet = lambda x, t, e: (0.5 * x).astype(np.float32)
    full = T.run(unet, S.make("lms"), ehs, init, 8, 3.0)
    part = T.run(unet, S.make("lms"), ehs, full["latents"][2], 8, 3.0, start_step=3)
    assert np.isfinite(part["final"]).all() and part["final"].shape == full["final"].shape
    assert not np.allclose(part["final"], full["final"], atol=1e-7)   # the history differs, so the trajectories do


def test_modelgen_vae_encoder_binds_to_the_restatement(tmp_path):
    from diffusers_amd import modelgen
    cfg = modelgen.vae_config(block_out_channels=(32, 64), layers_per_block=1)
    shapes = modelgen.vae_encoder_param_shapes(cfg)
    w = modelgen.random_weights(shapes, 9)
    path = str(tmp_path / "vae_encoder.onnx")
    modelgen.write_vae_encoder_onnx(path, cfg, w)
    m = onnx_io.load(path)
    bound = FM.bind_all(m)
    for name, shp in shapes.items():
        assert name in bound, name
        got = tuple(bound[name].shape)
        assert got == tuple(shp) or (len(shp) == 2 and got == tuple(shp)), (name, got, shp)
    ops = {n.op_type for n in m.nodes}
    assert {"Pad", "Conv", "Slice", "Clip", "Exp", "RandomNormalLike"} <= ops
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (1, 3, 24, 16)).astype(np.float32))
    with torch.no_grad():
        mean = FM.vae_encoder(bound, x, None)
    assert tuple(mean.shape) == (1, 4, 12, 8) and torch.isfinite(mean).all()
