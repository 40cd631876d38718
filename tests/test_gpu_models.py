"""GPU: native UNet / VAE decoder / txt2img loop (through the C ABI) against the oracle on identical synthetic weights,
and against the committed golden I/O of the reference's own VAE fixture graph.
Bars (BASELINE.json north_star): per-step noise prediction rel-L2 <= 1e-2 (fp16); final image PSNR >= 35 dB."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 10 * np.log10(1.0 / max(mse, 1e-20))


@pytest.fixture(scope="module")
def pipe(tiny_model_dir):
    from diffusers_amd.pipeline import StableDiffusionPipeline
    p = StableDiffusionPipeline(tiny_model_dir)
    yield p
    p.close()


@pytest.fixture(scope="module")
def pipe_plain(tiny_model_dir):
    from diffusers_amd.pipeline import StableDiffusionOptions, StableDiffusionPipeline
    p = StableDiffusionPipeline(tiny_model_dir, StableDiffusionOptions(force_naive=True))
    yield p
    p.close()


@pytest.fixture(scope="module")
def oracle_models(tiny_model_dir):
    from oracle import model_dir
    uw, ucfg = model_dir.load_unet(tiny_model_dir)
    vw, vcfg = model_dir.load_vae_decoder(tiny_model_dir)
    t = lambda d: {k: torch.from_numpy(v) for k, v in d.items()}
    return t(uw), ucfg, t(vw), vcfg


def test_loader_infers_architecture(pipe):
    mi = pipe.info()
    assert mi.unet_loaded and mi.vae_loaded
    assert list(mi.block_out_channels[:2]) == [64, 128] and mi.n_levels == 2 and mi.layers_per_block == 1
    assert list(mi.num_heads[:2]) == [4, 8] or list(mi.num_heads[:2]) == [4, 4]
    assert mi.cross_attention_dim == 64 and not mi.use_linear_projection
    assert list(mi.vae_block_out_channels[:2]) == [128, 64] and mi.vae_layers_per_block == 1


# odd latent sizes: the reference accepts every image size divisible by 8 (impl_txt2img.rs:251-253) and the exported graph
# upsamples to the skip connection's size (traced on a 64 x 65 latent, sd2pyke.py:287-288)
@pytest.mark.parametrize("N,h,w,S,t", [(2, 16, 16, 77, 946.42), (1, 8, 24, 152, 0.0), (4, 32, 32, 227, 999.0),
                                       (2, 15, 17, 77, 500.0), (1, 9, 9, 77, 20.0), (1, 1, 3, 77, 7.0)])
def test_unet_forward_vs_oracle(pipe, pipe_plain, oracle_models, N, h, w, S, t):
    from oracle import unet_ref
    uw, ucfg, _, _ = oracle_models
    rng = np.random.default_rng(N * 100 + h)
    x = rng.standard_normal((N, 4, h, w)).astype(np.float32)
    ehs = (rng.standard_normal((N, S, 64)) * 0.5).astype(np.float32)
    ref = unet_ref.unet_forward(uw, ucfg, torch.from_numpy(x), t, torch.from_numpy(ehs)).numpy()
    got = pipe.unet(x, t, ehs)
    plain = pipe_plain.unet(x, t, ehs)
    assert rel_l2(plain, ref) < 1e-2
    assert rel_l2(got, ref) < 1e-2
    assert rel_l2(got, plain) < 1e-2


def test_unet_vs_graph_interpreter(pipe, tiny_model_dir):
    """The engine against the SECOND route: the very unet.onnx it loaded, executed node by node by the ONNX interpreter
    (pinned on the reference's own exports, tests/test_oracle_fixture_graphs.py) instead of the functional restatement."""
    from oracle import onnx_interp, onnx_io
    m = onnx_io.load(os.path.join(tiny_model_dir, "unet.onnx"))
    rng = np.random.default_rng(404)
    for N, h, w, S, t in [(2, 16, 16, 77, 603.7), (1, 15, 17, 152, 41.0)]:
        x = rng.standard_normal((N, 4, h, w)).astype(np.float32)
        ehs = (rng.standard_normal((N, S, 64)) * 0.5).astype(np.float32)
        ref = onnx_interp.run(m, {"sample": x, "timestep": np.asarray([t], np.float32), "encoder_hidden_states": ehs})["out_sample"]
        assert rel_l2(pipe.unet(x, t, ehs), ref) < 1e-2


def test_timestep_truncation(pipe):
    """The exported graph casts timestep to int64 (sd2pyke.py:121): 946.42 and 946.0 must give identical outputs."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
    ehs = rng.standard_normal((2, 77, 64)).astype(np.float32)
    assert np.array_equal(pipe.unet(x, 946.42, ehs), pipe.unet(x, 946.0, ehs))
    assert not np.array_equal(pipe.unet(x, 946.42, ehs), pipe.unet(x, 945.0, ehs))


def test_vae_decode_vs_oracle(pipe, oracle_models):
    from oracle import unet_ref
    _, _, vw, vcfg = oracle_models
    rng = np.random.default_rng(2)
    lat = (rng.standard_normal((3, 4, 16, 16)) * 0.18215).astype(np.float32)
    ref = unet_ref.decode_latents(vw, vcfg, torch.from_numpy(lat)).numpy()
    got = pipe.decode_latents(lat)
    assert got.shape == ref.shape == (3, 32, 32, 3)
    assert got.min() >= 0.0 and got.max() <= 1.0
    assert psnr(got, ref) >= 35.0
    raw = pipe.vae_decoder(lat)
    raw_ref = unet_ref.vae_decode(vw, vcfg, torch.from_numpy(lat)).numpy()
    assert rel_l2(raw, raw_ref) < 1e-2


def test_reference_vae_fixture_graph():
    """The reference's own tests/stable-diffusion/vae_decoder.onnx (unaligned channels 32/64 -> plain kernels + flash attention)."""
    from diffusers_amd.pipeline import StableDiffusionPipeline
    g = np.load(os.path.join(GOLD, "vae_fixture_io.npz"))
    p = StableDiffusionPipeline(None, vae_decoder_onnx=os.path.join(GOLD, "vae_decoder.onnx"), load_unet=False)
    mi = p.info()
    assert list(mi.vae_block_out_channels[:2]) == [64, 32] and mi.vae_latent_channels == 4
    out = p.vae_decoder(g["latent"])
    assert rel_l2(out, g["sample"]) < 1e-2
    p.close()


def test_approximate_decode(pipe):
    rng = np.random.default_rng(3)
    lat = rng.standard_normal((2, 4, 8, 8)).astype(np.float32)
    coefs = np.array([[0.298, 0.207, 0.208], [0.187, 0.286, 0.173], [-0.158, 0.189, 0.264], [-0.184, -0.271, -0.473]], np.float32)
    ref = np.clip(np.einsum("blxy,lr->bxyr", lat, coefs), 0, 1)
    assert np.abs(pipe.approximate_decode_latents(lat) - ref).max() < 1e-5


@pytest.mark.parametrize("kind,steps,guidance,rescale", [("euler", 8, 7.5, None), ("euler_a", 6, 7.5, None), ("dpmpp", 8, 7.5, None),
                                                         ("ddim", 8, 7.5, 0.7), ("ddpm", 5, 1.0, None), ("lms", 8, 5.0, None)])
def test_txt2img_trace_vs_oracle(pipe, oracle_models, kind, steps, guidance, rescale):
    """Whole loop: same embeddings, same injected N(0,1) draws; per-step CFG-combined noise prediction and final image."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    from oracle import schedulers as S, txt2img as T, unet_ref
    uw, ucfg, vw, vcfg = oracle_models
    B, h, w = 2, 16, 16
    rng = np.random.default_rng(11)
    n_emb = 2 * B if guidance > 1.0 else B
    ehs = (rng.standard_normal((n_emb, 77, 64)) * 0.5).astype(np.float32)
    init = rng.standard_normal((B, 4, h, w)).astype(np.float32)
    noise = rng.standard_normal((steps + 2, B, 4, h, w)).astype(np.float32)
    unet = lambda x, t, e: unet_ref.unet_forward(uw, ucfg, torch.from_numpy(x), t, torch.from_numpy(e)).numpy()
    ref = T.run(unet, S.make(kind), ehs, init, steps, guidance, rescale, noise)
    opts = StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, guidance_scale=guidance, rescale_cfg=rescale, steps=steps,
                                         text_embeddings=ehs, batch_size=B, init_latents=init, step_noise=noise)
    sch = N.BY_NAME[kind].stable_diffusion_v1_optimized_default()
    res = opts.run(pipe, sch, trace=True)
    assert res.unet_calls == len(ref["timesteps"])
    for i in range(res.unet_calls):
        assert rel_l2(res.noise_pred[i], ref["noise_pred"][i]) < 1e-2, (kind, i)
    assert rel_l2(res.final_latents, ref["final"]) < 1e-2
    ref_img = unet_ref.decode_latents(vw, vcfg, torch.from_numpy(ref["final"])).numpy()
    assert res.images.shape == ref_img.shape          # the tiny VAE has 2 levels => 2x upscale (SD VAEs: 8x)
    assert psnr(res.images, ref_img) >= 35.0


def test_txt2img_odd_latent_size(pipe, oracle_models):
    """120 x 136 pixels (divisible by 8, not by 16): 15 x 17 latents through the whole loop and the decoder."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    from oracle import schedulers as S, txt2img as T, unet_ref
    uw, ucfg, vw, vcfg = oracle_models
    B, h, w, steps = 2, 15, 17, 4
    rng = np.random.default_rng(1517)
    ehs = (rng.standard_normal((2 * B, 77, 64)) * 0.5).astype(np.float32)
    init = rng.standard_normal((B, 4, h, w)).astype(np.float32)
    unet = lambda x, t, e: unet_ref.unet_forward(uw, ucfg, torch.from_numpy(x), t, torch.from_numpy(e)).numpy()
    ref = T.run(unet, S.make("dpmpp"), ehs, init, steps, 7.5)
    res = StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, steps=steps, text_embeddings=ehs, batch_size=B,
                                        init_latents=init).run(pipe, N.DPMSolverMultistepScheduler.stable_diffusion_v1_optimized_default(), trace=True)
    for i in range(res.unet_calls):
        assert rel_l2(res.noise_pred[i], ref["noise_pred"][i]) < 1e-2, i
    ref_img = unet_ref.decode_latents(vw, vcfg, torch.from_numpy(ref["final"])).numpy()
    assert res.images.shape == ref_img.shape and psnr(res.images, ref_img) >= 35.0


def test_txt2img_size_check_and_callbacks(pipe):
    from diffusers_amd import _lib, schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    ehs = np.zeros((2, 77, 64), np.float32)
    sch = N.EulerDiscreteScheduler.stable_diffusion_v1_optimized_default()
    bad = StableDiffusionTxt2ImgOptions(height=100, width=128, text_embeddings=ehs, batch_size=1, steps=2)
    with pytest.raises(_lib.SdeError, match="divisible by 8"):
        bad.run(pipe, sch, skip_decode=True)
    # an empty schedule (DDIM, one inference step: ddim.rs:184 yields n - 1 timesteps): the reference runs no step and decodes
    # init_noise_sigma * latents (impl_txt2img.rs:265,272,326), and so does the engine in parity mode; with the quirks off it is refused
    init = np.random.default_rng(9).standard_normal((1, 4, 16, 16)).astype(np.float32)
    one = StableDiffusionTxt2ImgOptions(height=128, width=128, text_embeddings=ehs, batch_size=1, steps=1, init_latents=init)
    r = one.run(pipe, N.BY_NAME["ddim"].stable_diffusion_v1_optimized_default(), trace=True)
    assert r.unet_calls == 0 and np.array_equal(r.final_latents, init)            # DDIM: init_noise_sigma = 1
    assert np.array_equal(r.images, pipe.decode_latents(init))
    seen = []
    o = StableDiffusionTxt2ImgOptions(height=128, width=128, text_embeddings=ehs, batch_size=1, steps=6, seed=1)
    o.callback_progress(2, lambda i, t: (seen.append(i) or True) and i < 2)   # stop after step 2
    r = o.run(pipe, sch, skip_decode=True)
    assert seen == [0, 2] and r.unet_calls == 3
    lat = []
    o = StableDiffusionTxt2ImgOptions(height=128, width=128, text_embeddings=ehs, batch_size=1, steps=3, seed=1)
    o.callback_latents(1, lambda i, t, l: lat.append(l.shape) or True)
    o.run(pipe, sch, skip_decode=True)
    assert lat == [(1, 4, 16, 16)] * 3


def _fires(nt, steps, kind, freq):
    """impl_txt2img.rs:270,306-323 restated: which step indices invoke the callback.  num_warmup_steps = len(timesteps) - steps
    is a usize: when it underflows (DDIM: n - 1 timesteps) it wraps in release builds and only `i == len - 1` can fire."""
    warm = nt - steps
    if warm < 0:
        warm = 1 << 62
    out = []
    for i in range(nt):
        if i == nt - 1 or (i + 1) > warm:
            if kind in ("progress", "latents"):
                if i % freq == 0:
                    out.append(i)
            elif i != 0 and i % freq == 0:
                out.append(i)
    return out


@pytest.mark.parametrize("sched,steps,freq", [("euler", 6, 1), ("euler", 6, 2), ("ddim", 6, 1), ("ddim", 7, 2), ("ddpm", 6, 1), ("dpmpp", 5, 3)])
@pytest.mark.parametrize("kind", ["latents", "decoded", "approximate"])
def test_in_loop_payload_callbacks(pipe, sched, steps, freq, kind):
    """StableDiffusionCallback::{Latents, Decoded, ApproximateDecoded} inside the loop (impl_txt2img.rs:306-323): the firing
    rule (frequency, `i != 0`, the DDIM usize wrap, DDPM's extra timestep) and the payload, which must be what
    decode_latents / approximate_decode_latents give for the latents that step left."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    B, h, w = 2, 16, 16
    rng = np.random.default_rng(77)
    ehs = (rng.standard_normal((2 * B, 77, 64)) * 0.5).astype(np.float32)
    sch = N.BY_NAME[sched].stable_diffusion_v1_optimized_default()
    seen = []
    o = StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, steps=steps, text_embeddings=ehs, batch_size=B, seed=9)
    getattr(o, "callback_" + kind)(freq, lambda i, t, a: seen.append((i, t, a)) or True)
    res = o.run(pipe, sch, trace=True)
    nt = res.unet_calls
    assert nt == len(sch.timesteps())
    assert [i for i, _, _ in seen] == _fires(nt, steps, kind, freq), (sched, kind)
    for i, t, a in seen:
        assert t == float(sch.timesteps()[i])
        lat = res.latents[i]
        if kind == "latents":
            assert np.array_equal(a, lat)
        elif kind == "decoded":
            assert a.shape == res.images.shape and psnr(a, pipe.decode_latents(lat)) >= 60.0
        else:
            assert np.abs(a - pipe.approximate_decode_latents(lat)).max() < 1e-6


@pytest.mark.parametrize("kind", ["latents", "decoded"])
def test_payload_callback_stops_the_loop(pipe, kind):
    """`false` -> break, and the latents of THAT step are decoded (impl_txt2img.rs:319-326), although the payload is delivered
    while the next UNet call is already in flight."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    ehs = (np.random.default_rng(5).standard_normal((2, 77, 64)) * 0.5).astype(np.float32)
    mk = lambda: StableDiffusionTxt2ImgOptions(height=128, width=128, steps=6, text_embeddings=ehs, batch_size=1, seed=4)
    full = mk().run(pipe, N.EulerDiscreteScheduler.stable_diffusion_v1_optimized_default(), trace=True)
    seen = []
    o = mk()
    getattr(o, "callback_" + kind)(1, lambda i, t, a: seen.append(i) or i < 2)
    part = o.run(pipe, N.EulerDiscreteScheduler.stable_diffusion_v1_optimized_default(), trace=True)
    first = 1 if kind == "decoded" else 0
    assert seen == list(range(first, 3)) and part.unet_calls == 3
    assert np.array_equal(part.final_latents, full.latents[2])
    assert psnr(part.images, pipe.decode_latents(full.latents[2])) >= 60.0


def test_callback_reentry_is_refused_and_exceptions_propagate(pipe):
    from diffusers_amd import _lib, schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    ehs = np.zeros((2, 77, 64), np.float32)
    sch = N.EulerDiscreteScheduler.stable_diffusion_v1_optimized_default()
    o = StableDiffusionTxt2ImgOptions(height=128, width=128, text_embeddings=ehs, batch_size=1, steps=3, seed=1)
    o.callback_latents(1, lambda i, t, l: pipe.approximate_decode_latents(l) is not None)   # calls back into the running engine
    with pytest.raises(_lib.SdeError, match="re-entered"):
        o.run(pipe, sch, skip_decode=True)
    o = StableDiffusionTxt2ImgOptions(height=128, width=128, text_embeddings=ehs, batch_size=1, steps=3, seed=1)

    def boom(i, t):
        raise KeyError("user code failed")
    o.callback_progress(1, boom)
    with pytest.raises(KeyError):
        o.run(pipe, sch, skip_decode=True)
    # the engine is usable afterwards
    assert np.isfinite(pipe.approximate_decode_latents(np.zeros((1, 4, 8, 8), np.float32))).all()


def test_ddpm_many_timesteps_trace_and_capacity(pipe):
    """DDPM yields ceil(T / (T // n)) timesteps (ddpm.rs:181): 167 for n = 150.  The per-timestep buffers are sized from the
    scheduler, and a buffer that is too short is refused before anything is written."""
    import ctypes as C
    from diffusers_amd import _lib, schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    ehs = (np.random.default_rng(1).standard_normal((1, 77, 64)) * 0.5).astype(np.float32)
    sch = N.DDPMScheduler.stable_diffusion_v1_optimized_default()
    o = StableDiffusionTxt2ImgOptions(height=64, width=64, guidance_scale=1.0, steps=150, text_embeddings=ehs, batch_size=1, seed=2)
    res = o.run(pipe, sch, trace=True, skip_decode=True)
    assert res.unet_calls == 167 == len(sch.timesteps()) and res.noise_pred.shape[0] == 167
    assert np.isfinite(res.latents).all() and np.array_equal(res.latents[-1], res.final_latents)
    # raw ABI: 150 rows stated, 167 needed
    L = _lib.lib()
    ro = _lib.Txt2ImgOpts()
    L.sde_txt2img_opts_default(C.byref(ro))
    ro.height = ro.width = 64; ro.guidance_scale = 1.0; ro.steps = 150; ro.batch_size = 1; ro.has_seed = 1; ro.skip_decode = 1
    ro.ehs, ro.S, ro.D = ehs.ctypes.data, 77, 64
    buf = np.zeros((150, 1, 4, 8, 8), np.float32)
    ro.latents_out, ro.trace_rows = buf.ctypes.data, 150
    assert L.sde_txt2img_run(pipe._h, sch._h, C.byref(ro), None, None) == 1 and b"167" in L.sde_last_error()
    assert not buf.any()
    o.step_noise = np.zeros((150, 1, 4, 8, 8), np.float32)
    with pytest.raises(_lib.SdeError, match="step_noise"):
        o.run(pipe, sch, skip_decode=True)


def test_two_threads_one_engine(pipe):
    """§8(b): concurrent calls on one engine from different host threads serialise and do not disturb each other --
    host-pointer UNet calls of different shapes against a txt2img run with CFG rescale on the same engine."""
    import threading
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    rng = np.random.default_rng(31)
    xs = [rng.standard_normal((n, 4, s, s)).astype(np.float32) for n, s in [(1, 8), (4, 32), (2, 16)]]
    es = [(rng.standard_normal((x.shape[0], 77, 64)) * 0.5).astype(np.float32) for x in xs]
    want = [pipe.unet(x, 500.0, e) for x, e in zip(xs, es)]
    ehs = (rng.standard_normal((4, 77, 64)) * 0.5).astype(np.float32)
    mk = lambda: StableDiffusionTxt2ImgOptions(height=128, width=128, rescale_cfg=0.7, steps=6, text_embeddings=ehs, batch_size=2, seed=3)
    want_run = mk().run(pipe, N.DDIMScheduler.stable_diffusion_v1_optimized_default(), trace=True).final_latents
    errs, got_run = [], []

    def unets():
        try:
            for _ in range(6):
                for x, e, w in zip(xs, es, want):
                    assert np.array_equal(pipe.unet(x, 500.0, e), w)
        except BaseException as exc:
            errs.append(exc)

    def runs():
        try:
            for _ in range(3):
                got_run.append(mk().run(pipe, N.DDIMScheduler.stable_diffusion_v1_optimized_default(), trace=True).final_latents)
        except BaseException as exc:
            errs.append(exc)

    th = [threading.Thread(target=unets), threading.Thread(target=runs), threading.Thread(target=unets)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs
    assert len(got_run) == 3 and all(np.array_equal(g, want_run) for g in got_run)


def test_seeded_runs_are_reproducible(pipe):
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    ehs = (np.random.default_rng(0).standard_normal((2, 77, 64)) * 0.5).astype(np.float32)
    def go(seed):
        sch = N.EulerAncestralDiscreteScheduler.stable_diffusion_v1_optimized_default()
        return StableDiffusionTxt2ImgOptions(height=128, width=128, text_embeddings=ehs, batch_size=1, steps=4, seed=seed).run(pipe, sch, trace=True).final_latents
    a, b, c = go(5), go(5), go(6)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


@pytest.mark.parametrize("kind", ["euler_a", "ddpm", "euler"])
def test_seeded_run_draws_the_reference_streams(pipe, kind):
    """`seed` alone must give what the reference's two StdRng streams give (impl_txt2img.rs:249,262,267): the same run with
    the ORACLE's draws injected (latents from seed, per-step scheduler noise from seed + 31337, consumed step after step)
    ends bit-identical.  Euler draws and discards a tensor per step in the reference; nothing depends on it."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionTxt2ImgOptions
    from oracle import rng as R
    B, h, w, steps, seed = 2, 16, 16, 4, 1234
    ehs = (np.random.default_rng(3).standard_normal((2 * B, 77, 64)) * 0.5).astype(np.float32)
    sch = N.BY_NAME[kind].stable_diffusion_v1_optimized_default()
    seeded = StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, steps=steps, text_embeddings=ehs, batch_size=B,
                                           seed=seed).run(pipe, sch, trace=True)
    init = R.random_using((B, 4, h, w), R.StdRng.seed_from_u64(seed))
    srng = R.StdRng.seed_from_u64(seed + 31337)
    nt = seeded.unet_calls
    noise = np.zeros((nt, B, 4, h, w), np.float32)
    for i in range(nt):
        # ddpm.rs:189-193,223: no draw when the POSITION of t in `timesteps` is 0, i.e. on the first step (REFERENCE_QUIRK);
        # euler_ancestral_discrete.rs:203: a draw on every step
        if kind == "euler_a" or (kind == "ddpm" and i > 0):
            noise[i] = R.random_using((B, 4, h, w), srng)
    sch2 = N.BY_NAME[kind].stable_diffusion_v1_optimized_default()
    injected = StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, steps=steps, text_embeddings=ehs, batch_size=B,
                                             init_latents=init, step_noise=noise).run(pipe, sch2, trace=True)
    assert np.array_equal(seeded.final_latents, injected.final_latents)
    assert np.array_equal(seeded.noise_pred[0], injected.noise_pred[0])


@pytest.mark.parametrize("kind,rescale,stream_ordered", [("euler_a", None, False), ("dpmpp", 0.7, False), ("euler_a", None, True), ("dpmpp", 0.7, True)])
def test_cfg_pair_split_matches_single_engine(tiny_model_dir, pipe, kind, rescale, stream_ordered):
    """SURVEY 8e / BASELINE config 3: two engines each run one half of the CFG batch and swap halves once per step through
    the sde_exchange_fn hook.  Here the two 'ranks' are two threads with two engines on the one GPU and the exchange is
    a device-to-device copy between them (tests/test_sharding_gloo.py covers the torch.distributed transport); the
    result must match the single-engine run, whose UNet sees both halves in one batch."""
    import threading
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionPipeline, StableDiffusionTxt2ImgOptions
    from diffusers_amd.sharding import CfgPair

    class ThreadPair(CfgPair):
        slots = [None, None]
        barrier = threading.Barrier(2)

        def __init__(self, half):
            super().__init__(None, None, half)
            self.stream_ordered = stream_ordered

        def gather(self, full):
            ThreadPair.slots[self.half] = full[self.half].clone()
            torch.cuda.synchronize()
            ThreadPair.barrier.wait(timeout=60)
            full[1 - self.half].copy_(ThreadPair.slots[1 - self.half])
            torch.cuda.synchronize()
            ThreadPair.barrier.wait(timeout=60)
            return full

        # sde_exchange_stream_fn: nothing synchronises a stream with the host.  Each engine's stream records "my half is ready", waits
        # for the peer's event, copies the peer's half, records "I have read it" and waits for the peer's -- the host threads only
        # rendezvous to hand the event objects over (what an all-gather enqueued on the stream does inside RCCL).
        ready = [None, None]
        taken = [None, None]

        def exchange_on_stream(self, ptr, half_elems, my_half, stream):
            from diffusers_amd.sharding import _DevicePtr
            assert my_half == self.half
            ext = torch.cuda.ExternalStream(int(stream))
            with torch.cuda.stream(ext):
                full = torch.as_tensor(_DevicePtr(ptr, 2 * half_elems), device="cuda").view(2, half_elems)
                ThreadPair.slots[self.half] = full[self.half]
                ev = torch.cuda.Event(); ev.record(ext)
                ThreadPair.ready[self.half] = ev
                ThreadPair.barrier.wait(timeout=60)
                ext.wait_event(ThreadPair.ready[1 - self.half])
                full[1 - self.half].copy_(ThreadPair.slots[1 - self.half])
                ev2 = torch.cuda.Event(); ev2.record(ext)
                ThreadPair.taken[self.half] = ev2
                ThreadPair.barrier.wait(timeout=60)
                ext.wait_event(ThreadPair.taken[1 - self.half])     # my next UNet call overwrites the half the peer is copying

    B, h, w, steps = 2, 16, 16, 5
    rng = np.random.default_rng(23)
    ehs = (rng.standard_normal((2 * B, 77, 64)) * 0.5).astype(np.float32)
    init = rng.standard_normal((B, 4, h, w)).astype(np.float32)
    noise = rng.standard_normal((steps + 2, B, 4, h, w)).astype(np.float32)
    mk = lambda: StableDiffusionTxt2ImgOptions(height=h * 8, width=w * 8, guidance_scale=7.5, rescale_cfg=rescale, steps=steps,
                                               text_embeddings=ehs, batch_size=B, init_latents=init, step_noise=noise)
    whole = mk().run(pipe, N.BY_NAME[kind].stable_diffusion_v1_optimized_default(), trace=True)
    peer = StableDiffusionPipeline(tiny_model_dir)
    out, errs = [None, None], []

    def rank(half, session):
        try:
            out[half] = mk().run(session, N.BY_NAME[kind].stable_diffusion_v1_optimized_default(), trace=True,
                                 cfg_pair=ThreadPair(half))
        except Exception as exc:
            errs.append(exc)
            ThreadPair.barrier.abort()

    th = [threading.Thread(target=rank, args=(0, pipe)), threading.Thread(target=rank, args=(1, peer))]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    peer.close()
    assert not errs, errs
    assert out[0] is not None and out[1] is not None
    assert np.array_equal(out[0].final_latents, out[1].final_latents)            # replicated scheduler step: identical
    for i in range(whole.unet_calls):
        assert rel_l2(out[0].noise_pred[i], whole.noise_pred[i]) < 2e-3, (kind, i)   # UNet batch B vs 2B: other tile shapes
    B0 = (B + 1) // 2                                                             # each engine decodes its share
    assert psnr(out[0].images[:B0], whole.images[:B0]) >= 45.0 and psnr(out[1].images[B0:], whole.images[B0:]) >= 45.0
    # misuse: half without hook / without CFG
    from diffusers_amd import _lib
    o = mk()
    o.guidance_scale = 1.0
    o.text_embeddings = ehs[:B]
    with pytest.raises(_lib.SdeError, match="cfg_half"):
        o.run(pipe, N.BY_NAME[kind].stable_diffusion_v1_optimized_default(), cfg_pair=ThreadPair(0))


def test_failed_graph_capture_falls_back_to_plain_launches(pipe, tiny_model_dir, monkeypatch):
    """A UNet call is captured into a hipGraph on its second use.  If anything throws while the stream is capturing, the engine
    ends the capture, turns graphs off and replays the call launch by launch: the run completes with identical results
    (fault injected with SDE_TEST_CAPTURE_FAULT, read at engine creation)."""
    from diffusers_amd import schedulers as N
    from diffusers_amd.pipeline import StableDiffusionPipeline, StableDiffusionTxt2ImgOptions
    rng = np.random.default_rng(77)
    ehs = (rng.standard_normal((2, 77, 64)) * 0.5).astype(np.float32)
    init = rng.standard_normal((1, 4, 16, 16)).astype(np.float32)
    opts = StableDiffusionTxt2ImgOptions(height=128, width=128, steps=4, text_embeddings=ehs, batch_size=1, init_latents=init)
    good = opts.run(pipe, N.DPMSolverMultistepScheduler.stable_diffusion_v1_optimized_default(), trace=True)
    assert good.unet_calls == 4 and good.unet_graph_launches == 3       # the first call of a run fills the text K/V cache, uncaptured
    monkeypatch.setenv("SDE_TEST_CAPTURE_FAULT", "1")
    p2 = StableDiffusionPipeline(tiny_model_dir)
    monkeypatch.delenv("SDE_TEST_CAPTURE_FAULT")
    try:
        for _ in range(2):                                                # the second run must not try (and fail) again
            r = opts.run(p2, N.DPMSolverMultistepScheduler.stable_diffusion_v1_optimized_default(), trace=True)
            assert r.unet_calls == 4 and r.unet_graph_launches == 0
            assert np.array_equal(r.noise_pred, good.noise_pred) and np.array_equal(r.images, good.images)
    finally:
        p2.close()
