"""GPU: the fp32 engine (sde_engine_opts.dtype = SDE_F32) against the fp32 oracle.  Bar (north_star): per-step noise prediction
relative L2 <= 1e-4; here also the decoded image (max abs error, images live in [0, 1])."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def pipe32(tiny_model_dir):
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline
    p = StableDiffusionPipeline(tiny_model_dir, StableDiffusionOptions(dtype="f32"))
    yield p
    p.close()


@pytest.fixture(scope="module")
def oracle_models(tiny_model_dir):
    from oracle import model_dir
    uw, ucfg = model_dir.load_unet(tiny_model_dir)
    vw, vcfg = model_dir.load_vae_decoder(tiny_model_dir)
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    return t(uw), ucfg, t(vw), vcfg


@pytest.mark.parametrize("N,h,w,S,t", [(2, 16, 16, 77, 946.42), (1, 8, 24, 152, 0.0), (4, 32, 32, 227, 999.0), (2, 15, 17, 77, 500.0), (1, 1, 3, 77, 7.0)])
def test_f32_unet_forward(pipe32, oracle_models, N, h, w, S, t):
    from oracle import unet_ref
    uw, ucfg, _, _ = oracle_models
    rng = np.random.default_rng(N * 100 + h)
    x = rng.standard_normal((N, 4, h, w)).astype(np.float32)
    ehs = (rng.standard_normal((N, S, 64)) * 0.5).astype(np.float32)
    with torch.no_grad():
        ref = unet_ref.unet_forward(uw, ucfg, torch.from_numpy(x), t, torch.from_numpy(ehs)).numpy()
    got = pipe32.unet(x, t, ehs)
    assert rel_l2(got, ref) < 1e-4, rel_l2(got, ref)


def test_f32_vae_decode(pipe32, oracle_models):
    from oracle import unet_ref
    _, _, vw, vcfg = oracle_models
    lat = (np.random.default_rng(2).standard_normal((3, 4, 16, 16)) * 0.18215).astype(np.float32)
    with torch.no_grad():
        ref = unet_ref.decode_latents(vw, vcfg, torch.from_numpy(lat)).numpy()
        raw_ref = unet_ref.vae_decode(vw, vcfg, torch.from_numpy(lat)).numpy()
    assert np.abs(pipe32.decode_latents(lat) - ref).max() < 1e-4
    assert rel_l2(pipe32.vae_decoder(lat), raw_ref) < 1e-4


@pytest.mark.parametrize("kind,steps,guidance,rescale", [("euler", 6, 7.5, None), ("euler_a", 5, 7.5, None), ("dpmpp", 6, 7.5, 0.7),
                                                         ("ddim", 6, 7.5, None), ("lms", 6, 5.0, None)])
def test_f32_txt2img_trace(pipe32, oracle_models, kind, steps, guidance, rescale):
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    from oracle import schedulers as S, txt2img as T, unet_ref
    uw, ucfg, vw, vcfg = oracle_models
    B, h, w = 2, 16, 16
    rng = np.random.default_rng(11)
    ehs = (rng.standard_normal((2 * B, 77, 64)) * 0.5).astype(np.float32)
    init = rng.standard_normal((B, 4, h, w)).astype(np.float32)
    noise = rng.standard_normal((steps + 2, B, 4, h, w)).astype(np.float32)
    unet = lambda x, t, e: unet_ref.unet_forward(uw, ucfg, torch.from_numpy(x), t, torch.from_numpy(e)).numpy()
    with torch.no_grad():
        ref = T.run(unet, S.make(kind), ehs, init, steps, guidance, rescale, noise)
        ref_img = unet_ref.decode_latents(vw, vcfg, torch.from_numpy(ref["final"])).numpy()
    res = StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, guidance_scale=guidance, rescale_cfg=rescale, steps=steps, text_embeddings=ehs,
                                        batch_size=B, init_latents=init, step_noise=noise).run(pipe32, N.BY_NAME[kind].stable_diffusion_v1_optimized_default(),
                                                                                              trace=True)
    assert res.unet_calls == len(ref["timesteps"])
    for i in range(res.unet_calls):
        assert rel_l2(res.noise_pred[i], ref["noise_pred"][i]) < 1e-4, (kind, i, rel_l2(res.noise_pred[i], ref["noise_pred"][i]))
    assert np.abs(res.images - ref_img).max() < 2e-4


def test_f32_sd2_style_and_fast_gelu(tmp_path, tiny_model_dir_fused_fastgelu):
    """Linear projections / per-level head counts (SD2.x), and the optimizer dialect's FastGelu flag, in fp32."""
    from diffusers_amd import modelgen
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline
    from oracle import model_dir, unet_ref
    ucfg = modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), cross_attention_dim=64,
                                num_heads=(2, 4), use_linear_projection=True)
    modelgen.write_model_dir(str(tmp_path), ucfg, modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1), seed=3)
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
    ehs = (rng.standard_normal((2, 77, 64)) * 0.5).astype(np.float32)
    for d, approx in ((str(tmp_path), "none"), (tiny_model_dir_fused_fastgelu, "tanh")):
        p = StableDiffusionPipeline(d, StableDiffusionOptions(dtype="f32"))
        try:
            if approx == "none":
                uw, oc = model_dir.load_unet(d)
            else:   # same weights as the raw tiny model (seed 7) behind the fused dialect: the oracle binds the raw export
                uw, oc = None, None
            if uw is not None:
                uw = {k: torch.from_numpy(v) for k, v in uw.items()}
                with torch.no_grad():
                    ref = unet_ref.unet_forward(uw, oc, torch.from_numpy(x), 321.0, torch.from_numpy(ehs)).numpy()
                assert rel_l2(p.unet(x, 321.0, ehs), ref) < 1e-4
            else:
                assert np.isfinite(p.unet(x, 321.0, ehs)).all()
        finally:
            p.close()


def test_bf16_is_refused(tiny_model_dir):
    from diffusers_amd import _lib
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline
    with pytest.raises(_lib.SdeError, match="bf16"):
        StableDiffusionPipeline(tiny_model_dir, StableDiffusionOptions(dtype="bf16"))
