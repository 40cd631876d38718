"""GPU: SD2.x-style geometry (BASELINE.json configs[3]) at reduced width: Linear proj_in/proj_out (`use_linear_projection`),
head dim 64 with a per-level head count, cross_attention_dim != block width, v-prediction with DDIM (reference quirk:
n-1 timesteps, src/schedulers/ddim.rs:184).  fp16 engine vs fp32 oracle; bars from BASELINE.json: per-step noise prediction
rel-L2 <= 1e-2, final image PSNR >= 35 dB."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 99.0 if mse == 0 else 10.0 * np.log10(1.0 / mse)


@pytest.fixture(scope="module")
def sd21(tmp_path_factory):
    from diffusers_amd import modelgen
    from diffusers_amd.pipeline import StableDiffusionPipeline
    from oracle import model_dir
    d = str(tmp_path_factory.mktemp("sd21_small"))
    ucfg = modelgen.unet_config(use_linear_projection=True, block_out_channels=(128, 256, 256), layers_per_block=1, down_attn=(True, True, False),
                                cross_attention_dim=192, num_heads=(2, 4, 4))
    vcfg = modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1)
    modelgen.write_model_dir(d, ucfg, vcfg, seed=21)
    pipe = StableDiffusionPipeline(d)
    uw, oc = model_dir.load_unet(d)
    vw, vc = model_dir.load_vae_decoder(d)
    yield (pipe, {k: torch.from_numpy(v) for k, v in uw.items()}, oc, {k: torch.from_numpy(v) for k, v in vw.items()}, vc)
    pipe.close()


def test_sd21_architecture_inferred(sd21):
    pipe, _, oc, _, _ = sd21
    mi = pipe.info()
    assert mi.use_linear_projection and mi.cross_attention_dim == 192
    assert list(mi.num_heads[:3]) == [2, 4, 4]          # head dim 64 at every level
    assert oc["use_linear_projection"]


@pytest.mark.parametrize("N,h,w,S", [(2, 24, 24, 77), (1, 16, 32, 152)])
def test_sd21_unet_forward(sd21, N, h, w, S):
    from oracle import unet_ref
    pipe, uw, oc, _, _ = sd21
    rng = np.random.default_rng(S + h)
    x = rng.standard_normal((N, 4, h, w)).astype(np.float32)
    ehs = (rng.standard_normal((N, S, 192)) * 0.5).astype(np.float32)
    with torch.no_grad():
        ref = unet_ref.unet_forward(uw, oc, torch.from_numpy(x), 601.0, torch.from_numpy(ehs)).numpy()
    got = pipe.unet(x, 601.0, ehs)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) < 1e-2


def test_sd21_ddim_v_prediction_trace(sd21):
    from diffusers_amd import schedulers as Nn
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    from oracle import schedulers as S, txt2img as T, unet_ref
    pipe, uw, oc, vw, vc = sd21
    B, h, w, steps = 2, 24, 24, 6
    rng = np.random.default_rng(4)
    ehs = (rng.standard_normal((2 * B, 77, 192)) * 0.5).astype(np.float32)
    init = rng.standard_normal((B, 4, h, w)).astype(np.float32)
    with torch.no_grad():
        unet = lambda x, t, e: unet_ref.unet_forward(uw, oc, torch.from_numpy(x), t, torch.from_numpy(e)).numpy()
        ref = T.run(unet, S.make("ddim", prediction_type="v_prediction"), ehs, init, steps, 4.0, None, None)
    opts = StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, guidance_scale=4.0, steps=steps, text_embeddings=ehs, batch_size=B,
                                         init_latents=init)
    res = opts.run(pipe, Nn.DDIMScheduler.stable_diffusion_v1_optimized_default(prediction_type="v_prediction"), trace=True)
    assert res.unet_calls == len(ref["timesteps"]) == steps - 1      # the reference's DDIM runs n-1 UNet calls
    # per-step bar on the SAME latents (north_star: "same seed, latents and weights"): the oracle UNet is re-run on the
    # engine's own input of every step (DDIM: scale_model_input is the identity, init_noise_sigma = 1)
    for i in range(res.unet_calls):
        x = init if i == 0 else res.latents[i - 1]
        with torch.no_grad():
            out = unet(np.concatenate([x, x], 0), float(ref["timesteps"][i]), ehs)
        eps = out[:B] + 4.0 * (out[B:] - out[:B])   # the CFG combine multiplies the UNet error by ~guidance: 4.0 keeps headroom
        assert rel_l2(res.noise_pred[i], eps) < 1e-2, i
    assert rel_l2(res.final_latents, ref["final"]) < 3e-2             # free-running trajectories, 5 steps of CFG 4
    with torch.no_grad():
        ref_img = unet_ref.decode_latents(vw, vc, torch.from_numpy(ref["final"])).numpy()
    assert psnr(res.images, ref_img) >= 35.0
