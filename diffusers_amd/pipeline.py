"""Host mirror of the reference's pipeline API for the txt2img hot path:

  ``StableDiffusionPipeline::new`` / ``decode_latents`` / ``approximate_decode_latents``
      (/root/reference/src/pipelines/stable_diffusion/impl_main.rs:109-181,385-413)
  ``StableDiffusionTxt2ImgOptions`` builder + ``run``
      (/root/reference/src/pipelines/stable_diffusion/impl_txt2img.rs:15-48,67-219,246-327)
  ``StableDiffusionCallback`` (/root/reference/src/pipelines/stable_diffusion/mod.rs:41-86)

The CLIP text encoder stays with the caller (it runs once per prompt on the reference's ORT path), so
``run`` takes the text embeddings ``encode_prompt`` would produce: f32 [2B,S,D] with the unconditional
half FIRST when guidance_scale > 1 (impl_main.rs:366-370), else [B,S,D].
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from . import _lib
from .schedulers import DiffusionScheduler


@dataclass
class StableDiffusionOptions:
    """pipelines/stable_diffusion/mod.rs:35-38 — here the only device choice is the MI355X ordinal."""
    device_id: int = 0
    dtype: str = "f16"          # engine precision: "f16" (MFMA kernels, benchmarked) or "f32" (the reference's own precision, fp32 MFMA tiles, ~8x slower)
    force_naive: bool = False
    num_heads_override: int = 0


class StableDiffusionPipeline:
    def __init__(self, root: str, options: Optional[StableDiffusionOptions] = None, *, unet_onnx: Optional[str] = None,
                 vae_decoder_onnx: Optional[str] = None, load_unet: bool = True, load_vae: bool = True):
        L = _lib.lib()
        options = options or StableDiffusionOptions()
        o = _lib.EngineOpts()
        L.sde_engine_opts_default(C.byref(o))
        o.device_id = options.device_id
        o.dtype = {"f16": 0, "bf16": 1, "f32": 2}[options.dtype]
        o.force_naive = int(options.force_naive)
        o.num_heads_override = options.num_heads_override
        o.load_unet, o.load_vae = int(load_unet), int(load_vae)
        h = C.c_void_p()
        if root is not None:
            _lib.check(L.sde_engine_create(root.encode(), C.byref(o), C.byref(h)))
        else:
            _lib.check(L.sde_engine_create_from_files(unet_onnx.encode() if unet_onnx else None,
                                                      vae_decoder_onnx.encode() if vae_decoder_onnx else None, C.byref(o), C.byref(h)))
        self._h = h
        self.options = options

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().sde_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown: the ctypes module may already be torn down
            pass

    def info(self) -> _lib.ModelInfo:
        mi = _lib.ModelInfo()
        _lib.check(_lib.lib().sde_engine_info(self._h, C.byref(mi)))
        return mi

    # --- the two ORT calls of the hot path
    def unet(self, sample, timestep: float, encoder_hidden_states) -> np.ndarray:
        """session.unet.run([sample, timestep, encoder_hidden_states]) (impl_txt2img.rs:283)."""
        x = np.ascontiguousarray(sample, np.float32)
        e = np.ascontiguousarray(encoder_hidden_states, np.float32)
        N, _, h, w = x.shape
        out = np.empty((N, self.info().out_channels, h, w), np.float32)
        _lib.check(_lib.lib().sde_unet_forward(self._h, x.ctypes.data, N, h, w, float(timestep), e.ctypes.data, e.shape[1], e.shape[2],
                                               out.ctypes.data))
        return out

    def vae_decoder(self, latent_sample) -> np.ndarray:
        """vae_decoder.run([latent_sample]) (impl_main.rs:403): f32 [B,3,8h,8w], no post-processing."""
        z = np.ascontiguousarray(latent_sample, np.float32)
        B, _, h, w = z.shape
        mi = self.info()
        up = 1 << (mi.vae_n_levels - 1)
        out = np.empty((B, mi.vae_out_channels, h * up, w * up), np.float32)
        _lib.check(_lib.lib().sde_vae_decoder_run(self._h, z.ctypes.data, B, h, w, out.ctypes.data))
        return out

    def load_vae_encoder(self, onnx_path: Optional[str] = None):
        """The `vae_encoder: Option<Session>` of impl_main.rs:58,140-150 (None: the model dir's `[vae] encoder` entry)."""
        _lib.check(_lib.lib().sde_engine_load_vae_encoder(self._h, onnx_path.encode() if onnx_path else None))
        return self

    def vae_encoder(self, sample, noise=None) -> np.ndarray:
        """vae_encoder.run([sample]) of the exported graph (sd2pyke.py:181-187): f32 [B,3,H,W] in [-1,1] -> latent_sample
        [B,4,H/8,W/8] = mean + std * noise (the graph's RandomNormalLike becomes the `noise` argument; None = posterior mean)."""
        x = np.ascontiguousarray(sample, np.float32)
        B, _, H, W = x.shape
        L = _lib.lib()
        lc, h, w = C.c_int(), C.c_int(), C.c_int()
        _lib.check(L.sde_vae_encoder_out_shape(self._h, H, W, C.byref(lc), C.byref(h), C.byref(w)))
        out = np.empty((B, lc.value, h.value, w.value), np.float32)
        nz = None
        if noise is not None:
            nz = np.ascontiguousarray(noise, np.float32)
            if nz.shape != out.shape:
                raise ValueError(f"noise must be {out.shape}, got {nz.shape}")
        _lib.check(L.sde_vae_encoder_run(self._h, x.ctypes.data, B, H, W, nz.ctypes.data if nz is not None else None, out.ctypes.data))
        return out

    def decode_latents(self, latents) -> np.ndarray:
        """impl_main.rs:398-413 → f32 [B,H,W,3] in [0,1] (the Rgb32F image buffers)."""
        z = np.ascontiguousarray(latents, np.float32)
        B, _, h, w = z.shape
        mi = self.info()
        up = 1 << (mi.vae_n_levels - 1)
        out = np.empty((B, h * up, w * up, mi.vae_out_channels), np.float32)
        _lib.check(_lib.lib().sde_vae_decode(self._h, z.ctypes.data, B, h, w, out.ctypes.data))
        return out

    def approximate_decode_latents(self, latents) -> np.ndarray:
        """impl_main.rs:385-395."""
        z = np.ascontiguousarray(latents, np.float32)
        B, _, h, w = z.shape
        out = np.empty((B, h, w, 3), np.float32)
        _lib.check(_lib.lib().sde_approximate_decode(self._h, z.ctypes.data, B, h, w, out.ctypes.data))
        return out

    # --- measurement hooks
    def profile(self, enable: bool):
        _lib.check(_lib.lib().sde_profile_enable(self._h, int(enable)))

    def profile_reset(self):
        _lib.check(_lib.lib().sde_profile_reset(self._h))

    def profile_get(self) -> List[dict]:
        L = _lib.lib()
        n = C.c_int()
        _lib.check(L.sde_profile_get(self._h, None, 0, C.byref(n)))          # class count first (per-shape tables exceed any fixed size)
        cap = max(n.value, 1)
        arr = (_lib.KernelStat * cap)()
        _lib.check(L.sde_profile_get(self._h, arr, cap, C.byref(n)))
        return [dict(name=arr[i].name.decode(), launches=int(arr[i].launches), total_ms=float(arr[i].total_ms),
                     flops=float(arr[i].flops), bytes=float(arr[i].bytes)) for i in range(min(n.value, cap))]


CB_NONE, CB_PROGRESS, CB_LATENTS, CB_DECODED, CB_APPROXIMATE = range(5)


@dataclass
class Txt2ImgResult:
    images: Optional[np.ndarray]           # f32 [B,H,W,3] in [0,1]
    unet_calls: int = 0
    total_ms: float = 0.0
    unet_ms: float = 0.0
    vae_ms: float = 0.0
    host_enqueue_ms: float = 0.0
    unet_graph_launches: int = 0
    noise_pred: Optional[np.ndarray] = None  # [n_steps,B,4,h,w] when trace=True
    latents: Optional[np.ndarray] = None     # [n_steps,B,4,h,w] when trace=True
    final_latents: Optional[np.ndarray] = None


@dataclass
class StableDiffusionTxt2ImgOptions:
    """Fields and defaults of impl_txt2img.rs:15-65.  ``positive_prompt``/``negative_prompt`` are replaced by the
    embeddings they encode to (``text_embeddings``) plus ``batch_size``."""
    height: int = 512
    width: int = 512
    guidance_scale: float = 7.5
    rescale_cfg: Optional[float] = None
    steps: int = 25
    seed: Optional[int] = None
    ensd: int = 0  # never read by the reference's run() (it uses seed + 31337, impl_txt2img.rs:267)
    text_embeddings: Optional[np.ndarray] = None
    batch_size: int = 1
    callback_kind: int = CB_NONE
    callback_frequency: int = 1
    callback: Optional[Callable] = None
    init_latents: Optional[np.ndarray] = None
    step_noise: Optional[np.ndarray] = None
    start_step: int = 0                       # image-to-image: run timesteps[start_step:] from already-noised init_latents

    # builder methods (impl_txt2img.rs:67-219)
    def with_size(self, width: int, height: int):
        return self.with_width(width).with_height(height)

    def with_width(self, width: int):
        self.width = max(width // 8, 1) * 8
        return self

    def with_height(self, height: int):
        self.height = max(height // 8, 1) * 8
        return self

    def with_steps(self, steps: int):
        self.steps = steps
        return self

    def with_text_embeddings(self, ehs, batch_size: int):
        self.text_embeddings = np.ascontiguousarray(ehs, np.float32)
        self.batch_size = batch_size
        return self

    def with_seed(self, seed: int):
        self.seed = seed
        return self

    def with_random_seed(self):
        self.seed = None
        return self

    def with_guidance_scale(self, g: float):
        self.guidance_scale = g
        return self

    def with_eta_noise_seed_delta(self, ensd: int):
        self.ensd = ensd
        return self

    # Callbacks run on the calling thread inside the engine's locked call (as the reference runs them inside `run`): they must not
    # call back into the same pipeline (unet / decode_latents / run ...); such a call raises SdeError("re-entered") instead of
    # deadlocking.  An exception raised by a callback stops the loop and is re-raised by run().
    def callback_progress(self, frequency: int, cb: Callable[[int, float], bool]):
        self.callback_kind, self.callback_frequency, self.callback = CB_PROGRESS, frequency, lambda i, t, p: cb(i, t)
        return self

    def callback_latents(self, frequency: int, cb):
        self.callback_kind, self.callback_frequency, self.callback = CB_LATENTS, frequency, cb
        return self

    def callback_decoded(self, frequency: int, cb):
        self.callback_kind, self.callback_frequency, self.callback = CB_DECODED, frequency, cb
        return self

    def callback_approximate(self, frequency: int, cb):
        self.callback_kind, self.callback_frequency, self.callback = CB_APPROXIMATE, frequency, cb
        return self

    def run(self, session: StableDiffusionPipeline, scheduler: DiffusionScheduler, *, trace: bool = False,
            skip_decode: bool = False, device_ptrs: Optional[dict] = None, cfg_pair=None) -> Txt2ImgResult:
        """impl_txt2img.rs:246-327.  ``device_ptrs`` (bench path) = dict(ehs=, init_latents=, step_noise=, images=)
        of raw device addresses; then nothing crosses PCIe inside the call.  ``cfg_pair`` (sharding.CfgPair): this
        process runs one half of the CFG batch and swaps halves with its peer once per step."""
        L = _lib.lib()
        o = _lib.Txt2ImgOpts()
        L.sde_txt2img_opts_default(C.byref(o))
        o.height, o.width, o.guidance_scale, o.steps = self.height, self.width, self.guidance_scale, self.steps
        if self.rescale_cfg is not None:
            o.has_rescale_cfg, o.rescale_cfg = 1, self.rescale_cfg
        if self.seed is not None:
            o.has_seed, o.seed = 1, self.seed
        B = self.batch_size
        o.batch_size = B
        h, w = self.height // 8, self.width // 8
        keep = []
        if device_ptrs is not None:
            o.buffers_on_device = 1
            o.ehs = device_ptrs["ehs"]
            o.S, o.D = device_ptrs["S"], device_ptrs["D"]
            o.init_latents = device_ptrs.get("init_latents")
            o.step_noise = device_ptrs.get("step_noise")
            o.step_noise_rows = device_ptrs.get("step_noise_rows", 0)
            images_ptr = device_ptrs.get("images")
            images = None
        else:
            if self.text_embeddings is None:
                raise ValueError("text_embeddings are required (CLIP stays on the caller's side)")
            ehs = np.ascontiguousarray(self.text_embeddings, np.float32)
            N = 2 * B if self.guidance_scale > 1.0 else B
            if ehs.ndim != 3 or ehs.shape[0] != N:
                raise ValueError(f"text_embeddings must be [{N},S,D] (uncond first under CFG), got {ehs.shape}")
            keep.append(ehs)
            o.ehs, o.S, o.D = ehs.ctypes.data, ehs.shape[1], ehs.shape[2]
            if self.init_latents is not None:
                il = np.ascontiguousarray(self.init_latents, np.float32)
                assert il.shape == (B, 4, h, w), il.shape
                keep.append(il)
                o.init_latents = il.ctypes.data
            if self.step_noise is not None:
                sn = np.ascontiguousarray(self.step_noise, np.float32)
                if sn.ndim < 2 or sn[0].size != B * 4 * h * w:
                    raise ValueError(f"step_noise must be [n_timesteps,{B},4,{h},{w}], got {sn.shape}")
                keep.append(sn)
                o.step_noise = sn.ctypes.data
                o.step_noise_rows = sn.shape[0]     # the engine rejects a run whose scheduler yields more timesteps than rows
            mi = session.info()
            up = 1 << (mi.vae_n_levels - 1) if mi.vae_loaded else 8   # 8 for SD VAEs => [B,H,W,3]
            images = None if skip_decode else np.empty((B, h * up, w * up, 3), np.float32)
            images_ptr = images.ctypes.data if images is not None else None
        o.skip_decode = int(skip_decode)
        o.start_step = int(self.start_step)
        res = Txt2ImgResult(images=images)
        if self.callback is not None and self.callback_kind != CB_NONE:
            kind = self.callback_kind
            user_cb = self.callback
            cb_mi = session.info()
            cb_up = 1 << (cb_mi.vae_n_levels - 1) if cb_mi.vae_loaded else 8

            cb_errors = []

            def tramp(step, t, payload, n, _user):
                try:
                    return tramp_body(step, t, payload, n)
                except BaseException as exc:   # ctypes would swallow it and read "stop": keep it and re-raise after the run
                    cb_errors.append(exc)
                    return 0

            def tramp_body(step, t, payload, n):
                arr = None
                if payload and n:
                    arr = np.ctypeslib.as_array(payload, shape=(n,)).copy()
                    if kind == CB_LATENTS:
                        arr = arr.reshape(B, 4, h, w)
                    elif kind == CB_DECODED:
                        arr = arr.reshape(B, h * cb_up, w * cb_up, 3)
                    elif kind == CB_APPROXIMATE:
                        arr = arr.reshape(B, h, w, 3)
                return 1 if user_cb(step, t, arr) else 0

            cfn = _lib.CALLBACK(tramp)
            keep.append(cfn)
            o.callback, o.callback_kind, o.callback_frequency = cfn, kind, self.callback_frequency
        else:
            cb_errors = []
        if trace and device_ptrs is None:
            # one row per TIMESTEP, which is the scheduler's count and not `steps` (DDIM: steps - 1; DDPM: ceil(T / (T // steps)),
            # e.g. 167 rows for steps = 150): run() calls set_timesteps(steps) itself (impl_txt2img.rs:264), doing it here first
            # is idempotent and tells us the size
            scheduler.set_timesteps(self.steps)
            nmax = len(scheduler.timesteps())
            o.trace_rows = nmax
            res.noise_pred = np.zeros((nmax, B, 4, h, w), np.float32)
            res.latents = np.zeros((nmax, B, 4, h, w), np.float32)
            res.final_latents = np.zeros((B, 4, h, w), np.float32)
            o.noise_pred_out, o.latents_out, o.final_latents_out = (res.noise_pred.ctypes.data, res.latents.ctypes.data,
                                                                    res.final_latents.ctypes.data)
        if cfg_pair is not None:
            pair_errors = []

            def xchg(ptr, half_elems, my_half, _user):
                try:
                    cfg_pair.exchange(ptr, half_elems, my_half)
                    return 1
                except Exception as exc:  # never let an exception cross the C boundary
                    pair_errors.append(exc)
                    return 0

            def xchg_stream(ptr, half_elems, my_half, stream, _user):
                try:
                    cfg_pair.exchange_on_stream(ptr, half_elems, my_half, stream)
                    return 1
                except Exception as exc:
                    pair_errors.append(exc)
                    return 0

            o.cfg_half = cfg_pair.half + 1
            if getattr(cfg_pair, "stream_ordered", False):      # the transport can be enqueued on the engine's stream: no per-step drain
                xsf = _lib.EXCHANGE_STREAM(xchg_stream)
                keep.append(xsf)
                o.exchange_stream = xsf
            else:
                xfn = _lib.EXCHANGE(xchg)
                keep.append(xfn)
                o.exchange = xfn
        st = _lib.RunStats()
        try:
            _lib.check(L.sde_txt2img_run(session._h, scheduler._h, C.byref(o), images_ptr, C.byref(st)))
        except _lib.SdeError as err:
            if cfg_pair is not None and pair_errors:
                raise pair_errors[0] from err
            raise
        if cb_errors:
            raise cb_errors[0]
        res.unet_calls, res.total_ms, res.unet_ms, res.vae_ms = st.unet_calls, st.total_ms, st.unet_ms, st.vae_ms
        res.host_enqueue_ms = st.host_enqueue_ms
        res.unet_graph_launches = st.unet_graph_launches
        if trace and res.noise_pred is not None:
            s0 = int(self.start_step)        # rows are indexed by the absolute step: the executed ones are [start_step, start_step + calls)
            res.noise_pred = res.noise_pred[s0:s0 + st.unet_calls]
            res.latents = res.latents[s0:s0 + st.unet_calls]
        return res


@dataclass
class StableDiffusionImg2ImgOptions:
    """Mirror of impl_img2img.rs:18-34 (fields, defaults, builder names).  The reference stops there: it has no `run` for these
    options, so `run` below follows diffusers 0.14.0's StableDiffusionImg2ImgPipeline, the pipeline pyke Diffusers tracks:
    image -> 2x - 1 -> VAE encoder posterior sample -> * 0.18215 -> scheduler.add_noise at timesteps[t_start] -> the txt2img loop
    over timesteps[t_start:], t_start = steps - min(int(steps * noise_strength), steps).  Randomness (no reference stream exists for
    it): the posterior noise, then the added noise, are consecutive draws of sde_rng_standard_normal(seed)."""
    reference_image: Optional[np.ndarray] = None      # f32 [B,3,H,W] in [0,1] (what with_image builds, :177-191)
    noise_strength: float = 0.6
    text_config: StableDiffusionTxt2ImgOptions = field(default_factory=StableDiffusionTxt2ImgOptions)

    def with_size(self, width: int, height: int):
        self.text_config.with_size(width, height)
        self.reference_image = None                    # :69-71 drop_reference_image
        return self

    def with_steps(self, steps: int):
        self.text_config.steps = steps
        return self

    def with_noise_strength(self, s: float):
        self.noise_strength = min(max(float(s), 0.0), 1.0)     # :141-144
        return self

    def with_seed(self, seed: int):
        self.text_config.seed = seed
        return self

    def with_guidance_scale(self, g: float):
        self.text_config.guidance_scale = g
        return self

    def with_text_embeddings(self, ehs, batch_size: int):
        self.text_config.with_text_embeddings(ehs, batch_size)
        return self

    def with_images(self, images_nhwc01):
        """:194-208 -- [B,H,W,3] in [0,1], already at the configured size (resizing stays with the caller's image library)."""
        x = np.ascontiguousarray(images_nhwc01, np.float32)
        if x.ndim != 4 or x.shape[3] != 3:
            raise ValueError(f"images must be [B,H,W,3], got {x.shape}")
        self.reference_image = np.ascontiguousarray(x.transpose(0, 3, 1, 2))
        return self

    def start_step(self) -> int:
        steps = self.text_config.steps
        return max(steps - min(int(steps * self.noise_strength), steps), 0)

    def run(self, session: StableDiffusionPipeline, scheduler: DiffusionScheduler, **kw) -> Txt2ImgResult:
        from . import _lib as L_
        tc = self.text_config
        if self.reference_image is None:
            raise ValueError("reference_image is not set")
        img = np.ascontiguousarray(self.reference_image, np.float32)
        B = img.shape[0]
        if B != tc.batch_size:
            raise ValueError("one reference image per prompt")
        seed = tc.seed if tc.seed is not None else int(np.random.SeedSequence().generate_state(1, np.uint64)[0])
        scheduler.set_timesteps(tc.steps)
        ts = scheduler.timesteps()
        t0 = self.start_step()
        lat_shape = (B, 4, tc.height // 8, tc.width // 8)
        n = int(np.prod(lat_shape))
        draws = np.empty(2 * n, np.float32)
        L_.check(L_.lib().sde_rng_standard_normal(C.c_uint64(seed), C.c_uint64(0), draws.ctypes.data, draws.size))
        lc, hh, ww = C.c_int(), C.c_int(), C.c_int()
        L_.check(L_.lib().sde_vae_encoder_out_shape(session._h, img.shape[2], img.shape[3], C.byref(lc), C.byref(hh), C.byref(ww)))
        if (lc.value, hh.value, ww.value) != lat_shape[1:]:
            raise ValueError(f"a {img.shape[2]}x{img.shape[3]} image encodes to {hh.value}x{ww.value} latents, the run is configured for "
                             f"{lat_shape[2]}x{lat_shape[3]} (height/8 x width/8)")
        z = session.vae_encoder(2.0 * img - 1.0, draws[:n].reshape(lat_shape))
        z = (np.float32(0.18215) * z).astype(np.float32)
        # timesteps[t0:] empty: nothing to denoise and no timestep to noise at, so the encoded image is decoded as it is.  That is
        # strength 0 for every scheduler -- and, for the reference's DDIM, whose schedule has steps - 1 entries (ddim.rs:184), also
        # every strength with int(steps * strength) == 1 (t0 == steps - 1 == len(ts)).  Parity unpinned (the reference has no img2img run).
        if t0 >= len(ts):
            return Txt2ImgResult(images=session.decode_latents(z), unet_calls=0)
        noised = scheduler.add_noise(z, draws[n:].reshape(lat_shape), float(ts[t0]))
        import copy
        cfg = copy.copy(tc)
        cfg.seed, cfg.init_latents, cfg.start_step = seed, noised, t0
        return cfg.run(session, scheduler, **kw)
