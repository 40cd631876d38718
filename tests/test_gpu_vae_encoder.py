"""GPU: the native VAE encoder (sde_engine_load_vae_encoder / sde_vae_encoder_run) and the image-to-image start of the loop.

The reference opens `vae_encoder.onnx` as a session (impl_main.rs:58,140-150) and never runs it (impl_img2img.rs has no `run`), so
what pins this path is the reference's own export: tests/golden/vae_encoder.onnx is the reference's fixture graph (data), and
tests/golden/vae_encoder_io.npz holds the outputs of that graph under the oracle's ONNX interpreter, itself checked against an
independent torch restatement (tests/test_oracle_fixture_graphs.py).  Bars: fp16 engine 1e-2 rel-L2, fp32 engine 1e-4."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


@pytest.fixture(scope="module")
def model_dir_enc(tmp_path_factory):
    from diffusers_amd import modelgen
    d = tmp_path_factory.mktemp("tiny_sd_enc")
    ucfg = modelgen.unet_config(block_out_channels=(64, 128), layers_per_block=1, down_attn=(True, False), cross_attention_dim=64, num_heads=4)
    vcfg = modelgen.vae_config(block_out_channels=(64, 128), layers_per_block=1)
    modelgen.write_model_dir(str(d), ucfg, vcfg, seed=7, vae_encoder=True)
    return str(d)


@pytest.mark.parametrize("dtype,bar", [("f16", 1e-2), ("f32", 1e-4)])
def test_reference_fixture_graph(dtype, bar):
    """The reference's real export (32/64-channel tiny VAE, 64 x 48 image): engine vs the interpreted graph, sample and mean."""
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline
    g = np.load(os.path.join(GOLD, "vae_encoder_io.npz"))
    pipe = StableDiffusionPipeline(None, StableDiffusionOptions(dtype=dtype), vae_decoder_onnx=os.path.join(GOLD, "vae_decoder.onnx"))
    try:
        with pytest.raises(Exception, match="no VAE encoder"):
            pipe.vae_encoder(g["sample"], g["noise"])
        pipe.load_vae_encoder(os.path.join(GOLD, "vae_encoder.onnx"))
        got = pipe.vae_encoder(g["sample"], g["noise"])
        assert got.shape == g["latent_sample"].shape
        assert rel_l2(got, g["latent_sample"]) < bar, rel_l2(got, g["latent_sample"])
        # posterior mean: the restatement with noise = None (the graph itself always samples)
        from oracle import fixture_models as FM, onnx_io
        w = FM.bind_all(onnx_io.load(os.path.join(GOLD, "vae_encoder.onnx")))
        with torch.no_grad():
            mean = FM.vae_encoder(w, torch.from_numpy(g["sample"]), None).numpy()
        assert rel_l2(pipe.vae_encoder(g["sample"]), mean) < bar
        with pytest.raises(Exception, match="no path given"):
            pipe.load_vae_encoder(None)          # created from files: there is no model dir to resolve the TOML entry against
    finally:
        pipe.close()


@pytest.mark.parametrize("B,H,W", [(2, 64, 48), (1, 40, 72), (3, 16, 16), (1, 34, 30)])
def test_tiny_encoder_mfma_kernels(model_dir_enc, B, H, W):
    """64/128-channel encoder (aligned widths: the MFMA conv kernels incl. the asymmetric-pad stride-2 window) vs the torch restatement."""
    from diffusers_amd.pipeline import StableDiffusionPipeline
    from oracle import fixture_models as FM, onnx_io
    pipe = StableDiffusionPipeline(model_dir_enc, load_unet=False).load_vae_encoder()      # path from the TOML's [vae] encoder
    try:
        w = FM.bind_all(onnx_io.load(os.path.join(model_dir_enc, "vae_encoder.onnx")))
        rng = np.random.default_rng(B * 1000 + H)
        x = rng.uniform(-1, 1, (B, 3, H, W)).astype(np.float32)
        with torch.no_grad():
            mean = FM.vae_encoder(w, torch.from_numpy(x), None).numpy()
            nz = rng.standard_normal(mean.shape).astype(np.float32)
            ref = FM.vae_encoder(w, torch.from_numpy(x), torch.from_numpy(nz)).numpy()
        assert rel_l2(pipe.vae_encoder(x), mean) < 1e-2
        assert rel_l2(pipe.vae_encoder(x, nz), ref) < 1e-2
    finally:
        pipe.close()


ASYM_CASES = [
    # NB, H, W, Cin, Cout -- pad (0,1,0,1) + 3x3 stride-2 pad-0 conv (Downsample2D(padding=0) of the VAE encoder)
    (2, 16, 16, 128, 128), (1, 15, 9, 64, 64), (1, 32, 32, 128, 256), (2, 64, 64, 128, 320), (4, 32, 32, 256, 160), (1, 7, 5, 24, 40),
]


def _kernel_takes(kern, case):
    # 0 heuristic, 1 plain: anything; 2 (256x320 / 256x256 tiles): Cout % 256 == 0 or % 320 == 0; 3 (256x160): Cout % 160 == 0; MFMA: Cin % 64 == 0
    _, _, _, Cin, Cout = case
    if kern == 2:
        return Cin % 64 == 0 and (Cout % 256 == 0 or Cout % 320 == 0)
    if kern == 3:
        return Cin % 64 == 0 and Cout % 160 == 0
    return True


@pytest.mark.parametrize("kern,case", [(k, c) for c in ASYM_CASES for k in (0, 1, 2, 3) if _kernel_takes(k, c)])
def test_asymmetric_pad_stride2_conv(model_dir_enc, case, kern):
    """Every conv kernel (heuristic, plain, 256x320 / 256x256, 256x160) against torch on fp16-rounded operands."""
    from diffusers_amd import _lib
    from diffusers_amd.pipeline import StableDiffusionPipeline
    NB, H, W, Cin, Cout = case
    pipe = StableDiffusionPipeline(model_dir_enc, load_unet=False)
    try:
        rng = np.random.default_rng(hash(case) % (2 ** 31))
        x = rng.standard_normal((NB, H, W, Cin)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        Ho, Wo = (H - 2) // 2 + 1, (W - 2) // 2 + 1
        out = np.empty((NB, Ho, Wo, Cout), np.float32)
        _lib.check(_lib.lib().sde_test_conv(pipe._h, kern, x.ctypes.data, NB, H, W, Cin, w.ctypes.data, b.ctypes.data, Cout, 3, 2, 3, out.ctypes.data, out.size))
        xt = F.pad(torch.from_numpy(x).half().float().permute(0, 3, 1, 2), (0, 1, 0, 1))
        ref = F.conv2d(xt, torch.from_numpy(w).half().float(), torch.from_numpy(b), stride=2, padding=0).permute(0, 2, 3, 1).numpy()
        assert out.shape == ref.shape
        assert rel_l2(out, ref) < 2e-3, rel_l2(out, ref)
    finally:
        pipe.close()


@pytest.mark.parametrize("kind,strength", [("euler", 0.6), ("euler_a", 0.5), ("ddim", 0.75), ("dpmpp", 0.3), ("lms", 0.6)])
def test_img2img_loop_start(model_dir_enc, kind, strength):
    """start_step: the loop over timesteps[t_start:] from noised image latents, engine vs the oracle loop (diffusers' img2img order:
    encode -> * 0.18215 -> add_noise at timesteps[t_start] -> denoise -> decode)."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionImg2ImgOptions, StableDiffusionPipeline
    from oracle import fixture_models as FM, model_dir, onnx_io, schedulers as S_, txt2img as T, unet_ref
    pipe = StableDiffusionPipeline(model_dir_enc).load_vae_encoder()
    try:
        B, H, W, steps, g = 2, 64, 48, 8, 5.0
        rng = np.random.default_rng(11)
        img = rng.uniform(0, 1, (B, H // 4, W // 4, 3)).astype(np.float32)     # the tiny VAE has two levels: x2, not x8
        ehs = (rng.standard_normal((2 * B, 77, 64)) * 0.5).astype(np.float32)
        opts = (StableDiffusionImg2ImgOptions().with_size(W, H).with_steps(steps).with_noise_strength(strength).with_seed(5)
                .with_guidance_scale(g).with_text_embeddings(ehs, B).with_images(img))
        t0 = opts.start_step()
        assert 0 < t0 < steps
        sched = N.BY_NAME[kind].stable_diffusion_v1_optimized_default()
        step_noise = rng.standard_normal((steps + 2, B, 4, H // 8, W // 8)).astype(np.float32)
        opts.text_config.step_noise = step_noise
        res = opts.run(pipe, sched, trace=True)

        # oracle: same draws (sde_rng_standard_normal is bit-identical to oracle/rng.py, tests/test_rng.py)
        from oracle import rng as ORNG
        n = B * 4 * (H // 8) * (W // 8)
        draws = np.asarray(ORNG.random_using((2 * n,), ORNG.StdRng.seed_from_u64(5)), np.float32)
        ew = FM.bind_all(onnx_io.load(os.path.join(model_dir_enc, "vae_encoder.onnx")))
        uw, ucfg = model_dir.load_unet(model_dir_enc)
        vw, vcfg = model_dir.load_vae_decoder(model_dir_enc)
        uw = {k: torch.from_numpy(v) for k, v in uw.items()}
        vw = {k: torch.from_numpy(v) for k, v in vw.items()}
        with torch.no_grad():
            x = torch.from_numpy(2.0 * img.transpose(0, 3, 1, 2) - 1.0)
            z = FM.vae_encoder(ew, x, torch.from_numpy(draws[:n].reshape(B, 4, H // 8, W // 8))).numpy() * np.float32(0.18215)
            osch = S_.make(kind)
            osch.set_timesteps(steps)
            noised = osch.add_noise(z.astype(np.float32), draws[n:].reshape(z.shape), osch.timesteps[t0])
            unet = lambda xx, t, e: unet_ref.unet_forward(uw, ucfg, torch.from_numpy(xx), t, torch.from_numpy(e)).numpy()
            ref = T.run(unet, S_.make(kind), ehs, noised, steps, g, None, step_noise, start_step=t0)
            ref_img = unet_ref.decode_latents(vw, vcfg, torch.from_numpy(ref["final"])).numpy()
        nt = len(ref["timesteps"])
        assert res.unet_calls == nt
        for j in range(nt):
            assert rel_l2(res.noise_pred[j], ref["noise_pred"][j]) < 1e-2, (kind, j)
        mse = float(np.mean((res.images - ref_img) ** 2))
        assert 10 * np.log10(1.0 / max(mse, 1e-20)) >= 35.0
        # misuse: start_step without latents / beyond the schedule
        from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
        bad = StableDiffusionTxt2ImgOptions(height=H, width=W, steps=steps, text_embeddings=ehs, batch_size=B, guidance_scale=g, start_step=2)
        with pytest.raises(Exception, match="needs init_latents"):
            bad.run(pipe, sched)
        bad.init_latents, bad.start_step = noised, 1000
        with pytest.raises(Exception, match="start_step"):
            bad.run(pipe, sched)
    finally:
        pipe.close()


def test_img2img_ddim_last_slot_is_empty(model_dir_enc):
    """The reference's DDIM yields steps - 1 timesteps (ddim.rs:184): a strength with int(steps * strength) == 1 gives
    t_start == steps - 1 == len(timesteps), an empty slice -- no UNet call, the encoded image is decoded un-noised (pipeline.py)."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionImg2ImgOptions, StableDiffusionPipeline
    pipe = StableDiffusionPipeline(model_dir_enc).load_vae_encoder()
    try:
        B, H, W, steps = 1, 64, 48, 8
        rng = np.random.default_rng(3)
        img = rng.uniform(0, 1, (B, H // 4, W // 4, 3)).astype(np.float32)
        ehs = (rng.standard_normal((2 * B, 77, 64)) * 0.5).astype(np.float32)
        opts = (StableDiffusionImg2ImgOptions().with_size(W, H).with_steps(steps).with_noise_strength(0.2).with_seed(5)
                .with_text_embeddings(ehs, B).with_images(img))
        sched = N.DDIMScheduler.stable_diffusion_v1_optimized_default()
        sched.set_timesteps(steps)
        assert opts.start_step() == steps - 1 == len(sched.timesteps())
        res = opts.run(pipe, sched)
        assert res.unet_calls == 0 and res.images.shape == (B, H // 4, W // 4, 3)
        # Euler at the same strength has a slot left: one UNet call
        res2 = opts.run(pipe, N.EulerDiscreteScheduler.stable_diffusion_v1_optimized_default())
        assert res2.unet_calls == 1
    finally:
        pipe.close()
