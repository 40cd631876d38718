"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

CPU restatement of ``StableDiffusionTxt2ImgOptions::run``
(/root/reference/src/pipelines/stable_diffusion/impl_txt2img.rs:246-327) in f32 numpy, with the RNG
draws injected (the reference's rand 0.8 StdRng / rand_distr StandardNormal stream cannot be reproduced
or pinned here — SURVEY.md Appendix D).  **parity unpinned** against the reference (no goldens there).
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

f32 = np.float32


def cfg_combine(noise_pred: np.ndarray, guidance: float, rescale_cfg: Optional[float]) -> np.ndarray:
    """impl_txt2img.rs:288-301 (uncond half first)."""
    n = noise_pred.shape[0] // 2
    u, t = noise_pred[:n], noise_pred[n:]
    x_cfg = (u + f32(guidance) * (t - u)).astype(np.float32)
    if rescale_cfg is None:
        return x_cfg
    ro_pos, ro_cfg = f32(t.std()), f32(x_cfg.std())   # scalar population std over the whole half batch
    x_res = (x_cfg * f32(ro_pos / ro_cfg)).astype(np.float32)
    m = f32(rescale_cfg)
    return (m * x_res + (f32(1.0) - m) * x_cfg).astype(np.float32)


def run(unet: Callable[[np.ndarray, float, np.ndarray], np.ndarray], scheduler, ehs: np.ndarray, init_noise: np.ndarray,
        steps: int, guidance: float = 7.5, rescale_cfg: Optional[float] = None, step_noise: Optional[np.ndarray] = None,
        start_step: int = 0):
    """Returns dict(timesteps, noise_pred [n,B,4,h,w], latents [n,B,4,h,w], final [B,4,h,w]).

    ``unet(x, t, ehs)`` plays session.unet.run; ``init_noise`` are the N(0,1) draws of :261-262;
    ``step_noise[i]`` the scheduler's draw at step i (Euler-a / DDPM).

    ``start_step`` > 0 is the image-to-image variant, which the reference does not have (impl_img2img.rs builds options only); it
    follows diffusers 0.14.0's StableDiffusionImg2ImgPipeline: ``init_noise`` are then the already-noised latents
    (scheduler.add_noise(0.18215 * encode(image), noise, timesteps[start_step])), taken as they are, and the loop runs
    timesteps[start_step:]; step_noise stays indexed by the absolute step."""
    do_cfg = guidance > 1.0                                   # :258
    scheduler.set_timesteps(steps)                            # :264
    latents = (init_noise.astype(np.float32) * f32(scheduler.init_noise_sigma())).astype(np.float32)  # :265
    if start_step > 0:
        latents = init_noise.astype(np.float32).copy()
    ts = np.array(scheduler.timesteps).copy()
    preds, lats = [], []
    for i, t in enumerate(ts):
        if i < start_step:
            continue
        x = np.concatenate([latents, latents], 0) if do_cfg else latents.copy()   # :273-277
        x = scheduler.scale_model_input(x, t)                                      # :278
        np_ = unet(x, float(t), ehs).astype(np.float32)                            # :283
        eps = cfg_combine(np_, guidance, rescale_cfg) if do_cfg else np_
        nz = step_noise[i] if (step_noise is not None and scheduler.needs_noise()) else None
        latents = scheduler.step(eps, t, latents, nz)                              # :303-304
        preds.append(eps)
        lats.append(latents.copy())
    # a schedule can be empty (the reference's DDIM with one inference step yields no timesteps): nothing runs, the latents stay
    empty = np.zeros((0,) + latents.shape, np.float32)
    return dict(timesteps=ts[start_step:], noise_pred=np.stack(preds) if preds else empty, latents=np.stack(lats) if lats else empty, final=latents)
