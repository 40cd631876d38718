"""ORACLE / TEST INFRASTRUCTURE — not part of the shipped product path.

CPU (numpy, strict float32) restatement of the reference's diffusion schedulers,
bug-for-bug.  Every function cites the reference file:line it follows
(paths relative to /root/reference).

Parity status: **parity unpinned** — the reference's own tests hold no numerical
golden for any scheduler; the reference is Rust and cannot be
built in this image (no cargo/rustc).  This restatement is pinned only by
(a) closed-form identities checked in tests/test_oracle_schedulers.py and
(b) the sanity values listed in SURVEY.md §3.3.

Third-party semantics restated here (not under /root/reference):
  * ndarray 0.15 ``Array1::linspace(a,b,n)``: ``a + i*((b-a)/(n-1))`` in f32
  * ndarray 0.15 ``Array1::range(a,b,s)``: ``ceil((b-a)/s)`` items ``a + i*s``
  * ndarray 0.15 ``.std(0.)``: population standard deviation
"""
from __future__ import annotations

import math
from typing import List, Optional

import numpy as np

f32 = np.float32

# --------------------------------------------------------------------------- ndarray helpers


def nd_linspace(a, b, n: int) -> np.ndarray:
    a, b = f32(a), f32(b)
    step = f32((b - a) / f32(n - 1)) if n > 1 else f32(0.0)
    i = np.arange(n, dtype=np.float32)
    return (a + step * i).astype(np.float32)


def nd_range(start, end, step) -> np.ndarray:
    start, end, step = f32(start), f32(end), f32(step)
    n = int(math.ceil(float(f32((end - start) / step))))
    n = max(n, 0)
    i = np.arange(n, dtype=np.float32)
    return (start + step * i).astype(np.float32)


# --------------------------------------------------------------------------- shared constructor math


def betas_for_alpha_bar(T: int, max_beta: float) -> np.ndarray:
    """schedulers/mod.rs:70-79 — note the *integer* divisions ``i / T`` (REFERENCE_QUIRK)."""

    def alpha_bar(ts: int):
        x = f32(f32(f32(ts) + f32(0.008)) / f32(1.008)) * f32(math.pi / 2)
        c = f32(np.cos(f32(x)))
        return f32(c * c)

    out = []
    for i in range(T):
        t1 = i // T
        t2 = (i + 1) // T
        out.append(min(f32(f32(1.0) - f32(alpha_bar(t2) / alpha_bar(t1))), f32(max_beta)))
    return np.asarray(out, dtype=np.float32)


def make_betas(T: int, beta_start: float, beta_end: float, schedule: str) -> np.ndarray:
    """euler_discrete.rs:74-84 (identical in all five schedulers)."""
    if schedule == "linear":
        return nd_linspace(beta_start, beta_end, T)
    if schedule == "scaled_linear":
        b = nd_linspace(np.sqrt(f32(beta_start)), np.sqrt(f32(beta_end)), T)
        return (b * b).astype(np.float32)
    if schedule == "squaredcos_cap_v2":
        return betas_for_alpha_bar(T, 0.999)
    raise ValueError(schedule)


def alphas_cumprod_of(betas: np.ndarray) -> np.ndarray:
    """euler_discrete.rs:86-95 — sequential f32 scan."""
    alphas = (f32(1.0) - betas).astype(np.float32)
    out = np.empty_like(alphas)
    p = f32(1.0)
    for i, a in enumerate(alphas):
        p = f32(p * a)
        out[i] = p
    return out


def _check_ctor(T, b0, b1):
    """euler_discrete.rs:64-72 error conditions."""
    if T == 0:
        raise ValueError("num_train_timesteps must be >0")
    for b in (b0, b1):
        fb = f32(b)
        if not np.isfinite(fb) or fb == 0 or abs(fb) < np.finfo(np.float32).tiny:
            raise ValueError("beta_start and beta_end must be normal")
    if f32(b0) >= f32(b1):
        raise ValueError("beta_start must be < beta_end")


# --------------------------------------------------------------------------- util/interpolation.rs


class LinearInterpolatorAccelerated:
    """util/interpolation.rs:42-105 (GSL-style cached bsearch)."""

    def __init__(self, x: np.ndarray, y: np.ndarray):
        self.x, self.y, self.cache = x, y, 0

    @staticmethod
    def _bsearch(xarr, x, lo, hi):
        while hi > lo + 1:
            i = (hi + lo) // 2
            if xarr[i] > x:
                hi = i
            else:
                lo = i
        return lo

    def _find(self, x):
        xi = self.cache
        if x < self.x[xi]:
            self.cache = self._bsearch(self.x, x, 0, xi)
        elif x >= self.x[xi + 1]:
            self.cache = self._bsearch(self.x, x, xi, len(self.x) - 1)
        return self.cache

    def eval(self, x) -> np.float32:
        x = f32(x)
        if x < self.x[0] or x > self.x[-1]:
            return f32(np.nan)
        idx = self._find(x)
        x_l, x_h = self.x[idx], self.x[idx + 1]
        y_l, y_h = self.y[idx], self.y[idx + 1]
        dx = f32(x_h - x_l)
        if dx > 0:
            return f32(y_l + f32(f32(f32(x - x_l) / dx) * f32(y_h - y_l)))
        return f32(np.nan)


# --------------------------------------------------------------------------- base


class _Base:
    kind = "?"
    prediction_type = "epsilon"
    # the Option<> fields of SchedulerStepOutput (mod.rs:109-150) the last `step` filled; DPM-Solver fills none (:379)
    last_step = dict(pred_original_sample=None, derivative=None)

    def order(self) -> int:
        return 1

    def init_noise_sigma(self) -> np.float32:
        return self._init_noise_sigma

    def __len__(self):
        return self.T

    def needs_noise(self) -> bool:
        return False


def _position(arr, v) -> int:
    """``.iter().position(|&p| p == timestep).unwrap()`` (euler_discrete.rs:135-140)."""
    idx = np.nonzero(arr == v)[0]
    if len(idx) == 0:
        raise KeyError(f"timestep out of this schedulers bounds: {v}")
    return int(idx[0])


# --------------------------------------------------------------------------- Euler / Euler-a


class EulerDiscrete(_Base):
    """schedulers/euler_discrete.rs (entire file)."""

    kind = "euler"

    def __init__(self, T=1000, beta_start=0.00085, beta_end=0.012, schedule="scaled_linear"):
        _check_ctor(T, beta_start, beta_end)
        self.T = T
        self.alphas_cumprod = alphas_cumprod_of(make_betas(T, beta_start, beta_end, schedule))
        ac = self.alphas_cumprod
        sig = np.sqrt(((f32(1.0) - ac) / ac).astype(np.float32)).astype(np.float32)
        self.sigmas = np.concatenate([sig[::-1], np.zeros(1, np.float32)])  # :101
        self.timesteps = nd_linspace(T - 1, 0.0, T)  # :103
        self._init_noise_sigma = f32(self.sigmas.max())  # :106-109 (never refreshed — REFERENCE_QUIRK)
        self.scale_called = False

    def set_timesteps(self, n: int):
        """:152-173"""
        ts = nd_linspace(self.T - 1, 0.0, n)
        ac = self.alphas_cumprod
        sig = np.sqrt(((f32(1.0) - ac) / ac).astype(np.float32)).astype(np.float32)
        xa = nd_range(0.0, float(len(sig)), 1.0)
        interp = LinearInterpolatorAccelerated(xa, sig)
        out = np.zeros(n + 1, np.float32)
        for i, x in enumerate(ts):
            out[i] = interp.eval(x)
        out[n] = 0.0
        self.sigmas, self.timesteps = out, ts

    def scale_model_input(self, sample: np.ndarray, t) -> np.ndarray:
        """:134-150"""
        i = _position(self.timesteps, f32(t))
        s = self.sigmas[i]
        self.scale_called = True
        return (sample / np.sqrt(f32(f32(s * s) + f32(1.0)))).astype(np.float32)

    def step(self, model_output, t, sample, noise=None) -> np.ndarray:
        """:175-221 with s_churn=0 ⇒ gamma=0.  The reference draws (and discards) a full
        N(0,1) tensor here (:200) — REFERENCE_QUIRK, affects only the RNG stream."""
        assert self.scale_called
        i = _position(self.timesteps, f32(t))
        sigma = self.sigmas[i]
        sigma_hat = f32(sigma * f32(1.0))
        pred = (sample - sigma_hat * model_output).astype(np.float32)
        deriv = ((sample - pred) / sigma_hat).astype(np.float32)
        dt = f32(self.sigmas[i + 1] - sigma_hat)
        self.last_step = dict(pred_original_sample=pred, derivative=deriv)   # :216-220 returns Some(pred_original_sample)
        return (sample + deriv * dt).astype(np.float32)

    def add_noise(self, x0, noise, t):
        """:223-241"""
        i = _position(self.timesteps, f32(t))
        return (x0 + noise * self.sigmas[i]).astype(np.float32)


class EulerAncestralDiscrete(EulerDiscrete):
    """schedulers/euler_ancestral_discrete.rs (constructor/set_timesteps identical to Euler)."""

    kind = "euler_a"

    def needs_noise(self):
        return True

    def step(self, model_output, t, sample, noise=None) -> np.ndarray:
        """:174-211"""
        assert self.scale_called
        i = _position(self.timesteps, f32(t))
        sf, st = self.sigmas[i], self.sigmas[i + 1]
        pred = (sample - sf * model_output).astype(np.float32)
        sf2, st2 = f32(sf * sf), f32(st * st)
        sigma_up = f32(np.sqrt(f32(f32(st2 * f32(sf2 - st2)) / sf2)))
        sigma_down = f32(np.sqrt(f32(st2 - f32(sigma_up * sigma_up))))
        deriv = ((sample - pred) / sf).astype(np.float32)
        dt = f32(sigma_down - sf)
        prev = (sample + deriv * dt).astype(np.float32)
        self.last_step = dict(pred_original_sample=pred, derivative=deriv)   # :206-210
        return (prev + noise.astype(np.float32) * sigma_up).astype(np.float32)


# --------------------------------------------------------------------------- DPM-Solver(++) multistep


class DPMSolverMultistep(_Base):
    """schedulers/dpm_solver_multistep.rs.

    REFERENCE_QUIRK (``reference_quirks=True``, the parity default): ``lower_order_nums`` is
    never incremented (:118,206,347,369) so *every* step takes the first-order update.
    ``reference_quirks=False`` gives the multistep solver the code describes (second order :259-285,
    third order :288-326) -- a named, separate mode, never the parity default (SURVEY.md §3.3 decision).
    ``algorithm_type`` ("dpmsolver++" | "dpmsolver", :30-36) and ``solver_type`` ("midpoint" | "heun",
    :43-49) follow the config struct (:71-74); ``algorithm_type`` is live in the quirk mode too
    (convert_model_output :232-244, first-order formula :255).
    """

    kind = "dpmpp"

    def __init__(self, T=1000, beta_start=0.00085, beta_end=0.012, schedule="scaled_linear",
                 prediction_type="epsilon", solver_order=2, reference_quirks=True,
                 algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True):
        _check_ctor(T, beta_start, beta_end)
        self.T = T
        self.prediction_type = prediction_type
        self.solver_order = solver_order
        self.reference_quirks = reference_quirks
        self.algorithm_type, self.solver_type, self.lower_order_final = algorithm_type, solver_type, lower_order_final
        ac = self.alphas_cumprod = alphas_cumprod_of(make_betas(T, beta_start, beta_end, schedule))
        self.alpha_t = np.sqrt(ac).astype(np.float32)
        self.sigma_t = np.sqrt((f32(1.0) - ac).astype(np.float32)).astype(np.float32)
        self.lambda_t = (np.log(self.alpha_t) - np.log(self.sigma_t)).astype(np.float32)
        self.timesteps = nd_linspace(T - 1, 0.0, T).astype(np.int64)  # `as usize` truncation (:190)
        self._init_noise_sigma = f32(1.0)
        self.model_outputs: List[np.ndarray] = []
        self.lower_order_nums = 0

    def set_timesteps(self, n: int):
        """:340-348"""
        self.timesteps = nd_linspace(self.T - 1, 0.0, n).astype(np.int64)
        self.model_outputs = []
        self.lower_order_nums = 0

    def scale_model_input(self, sample, t):
        return sample.astype(np.float32).copy()

    def convert_model_output(self, mo, t: int, sample):
        """:211-246"""
        a, s = self.alpha_t[t], self.sigma_t[t]
        if self.algorithm_type == "dpmsolver++":
            if self.prediction_type == "epsilon":
                return ((sample - s * mo) / a).astype(np.float32)
            if self.prediction_type == "sample":
                return mo.astype(np.float32).copy()
            return (a * sample - s * mo).astype(np.float32)  # v_prediction
        if self.prediction_type == "epsilon":
            return mo.astype(np.float32).copy()
        if self.prediction_type == "sample":
            return ((sample - a * mo) / s).astype(np.float32)
        return (a * mo + s * sample).astype(np.float32)

    def _exp_term(self, h):
        """exp(-h) - 1 (D++) or exp(h) - 1 (DPMSolver), as every update formula spells it"""
        if self.algorithm_type == "dpmsolver++":
            return f32(f32(np.exp(f32(-h))) - f32(1.0))
        return f32(f32(np.exp(f32(h))) - f32(1.0))

    def first_order_coeffs(self, t: int, prev_t: int):
        """:248-257 → (c_sample, c_m0, h) with prev = c_sample*sample - c_m0*m0"""
        lam_t, lam_s = self.lambda_t[prev_t], self.lambda_t[t]
        h = f32(lam_t - lam_s)
        ex = self._exp_term(h)
        if self.algorithm_type == "dpmsolver++":
            return f32(self.sigma_t[prev_t] / self.sigma_t[t]), f32(self.alpha_t[prev_t] * ex), h
        return f32(self.alpha_t[prev_t] / self.alpha_t[t]), f32(self.sigma_t[prev_t] * ex), h

    def step(self, model_output, t, sample, noise=None):
        """:350-380"""
        t = int(t)
        n = len(self.timesteps)
        i = _position(self.timesteps, t)
        last = i == n - 1
        prev_t = 0 if last else int(self.timesteps[i + 1])
        lower_final = last and self.lower_order_final and n < 15
        lower_second = (i == n - 2) and self.lower_order_final and n < 15
        m0 = self.convert_model_output(model_output, t, sample)
        self.model_outputs.append(m0)  # only ever pushed (:363-367) — grows; harmless here
        cs, c0, h = self.first_order_coeffs(t, prev_t)
        pp = self.algorithm_type == "dpmsolver++"
        outer = self.alpha_t[prev_t] if pp else self.sigma_t[prev_t]
        ex = self._exp_term(h)
        first = self.solver_order == 1 or self.lower_order_nums < 1 or lower_final
        if not self.reference_quirks and h == 0:
            # "fixed" mode: the final timestep 0 has prev_timestep 0, h == 0, and the multistep formulas divide by h;
            # the first-order update is the identity there (not a reference behaviour: the reference never gets here)
            first = True
        if first:
            prev = (cs * sample - c0 * m0).astype(np.float32)
        elif self.solver_order == 2 or self.lower_order_nums < 2 or lower_second:
            # :259-285; only reachable with reference_quirks=False
            s1 = int(self.timesteps[i - 1])
            m1 = self.model_outputs[-2]
            h0 = f32(self.lambda_t[t] - self.lambda_t[s1])
            r0 = f32(h0 / h)
            d1 = (f32(f32(1.0) / r0) * (m0 - m1)).astype(np.float32)
            if self.solver_type == "midpoint":
                prev = (cs * sample - c0 * m0 - f32(f32(0.5) * c0) * d1).astype(np.float32)
            elif pp:
                prev = (cs * sample - c0 * m0 + f32(outer * f32(f32(ex / h) + f32(1.0))) * d1).astype(np.float32)
            else:
                prev = (cs * sample - c0 * m0 - f32(outer * f32(f32(ex / h) - f32(1.0))) * d1).astype(np.float32)
        else:
            # :288-326
            s1, s2 = int(self.timesteps[i - 1]), int(self.timesteps[i - 2])
            m1, m2 = self.model_outputs[-2], self.model_outputs[-3]
            h0 = f32(self.lambda_t[t] - self.lambda_t[s1])
            h1 = f32(self.lambda_t[s1] - self.lambda_t[s2])
            r0, r1 = f32(h0 / h), f32(h1 / h)
            d10 = (f32(f32(1.0) / r0) * (m0 - m1)).astype(np.float32)
            d11 = (f32(f32(1.0) / r1) * (m1 - m2)).astype(np.float32)
            d1 = (d10 + f32(r0 / f32(r0 + r1)) * (d10 - d11)).astype(np.float32)
            d2 = (f32(f32(1.0) / f32(r0 + r1)) * (d10 - d11)).astype(np.float32)
            h2 = f32(h * h)
            if pp:
                c1 = f32(outer * f32(f32(ex / h) + f32(1.0)))
                c2 = f32(outer * f32(f32(f32(ex + h) / h2) - f32(0.5)))
                prev = (cs * sample - c0 * m0 + c1 * d1 - c2 * d2).astype(np.float32)
            else:
                c1 = f32(outer * f32(f32(ex / h) - f32(1.0)))
                c2 = f32(outer * f32(f32(f32(ex - h) / h2) - f32(0.5)))
                prev = (cs * sample - c0 * m0 - c1 * d1 - c2 * d2).astype(np.float32)
        if not self.reference_quirks and self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        return prev

    def add_noise(self, x0, noise, t):
        """:382-384"""
        ac = self.alphas_cumprod[int(t)]
        return (np.sqrt(ac) * x0 + np.sqrt(f32(f32(1.0) - ac)) * noise).astype(np.float32)


# --------------------------------------------------------------------------- DDIM


class DDIM(_Base):
    """schedulers/ddim.rs.  REFERENCE_QUIRK: ``set_timesteps(n)`` yields n-1 timesteps (:184)."""

    kind = "ddim"

    def __init__(self, T=1000, beta_start=0.00085, beta_end=0.012, schedule="scaled_linear",
                 prediction_type="epsilon", clip_sample=False, set_alpha_to_one=False, steps_offset=1):
        _check_ctor(T, beta_start, beta_end)
        self.T = T
        self.prediction_type = prediction_type
        self.clip_sample, self.steps_offset = clip_sample, steps_offset
        ac = self.alphas_cumprod = alphas_cumprod_of(make_betas(T, beta_start, beta_end, schedule))
        self.final_alpha_cumprod = f32(1.0) if set_alpha_to_one else ac[0]  # :132
        self.timesteps = nd_linspace(0.0, T - 1, T)[::-1].astype(np.int64)  # :134-137
        self._init_noise_sigma = f32(1.0)
        self.num_inference_steps: Optional[int] = None

    def set_timesteps(self, n: int):
        """:179-190"""
        self.num_inference_steps = n
        ratio = self.T // n
        ts = nd_range(0.0, float(n - 1), 1.0)[::-1]
        ts = np.round(ts * f32(ratio)).astype(np.int64)  # f32::round = half away from zero; exact ints here
        self.timesteps = (ts + self.steps_offset).astype(np.int64)

    def scale_model_input(self, sample, t):
        return sample.astype(np.float32).copy()

    def coeffs(self, t: int):
        prev_t = t - self.T // self.num_inference_steps  # :197
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, model_output, t, sample, noise=None):
        """:192-251 with ETA = 0"""
        t = int(t)
        a_t, a_prev = self.coeffs(t)
        b_t = f32(f32(1.0) - a_t)
        mo = model_output.astype(np.float32)
        if self.prediction_type == "epsilon":
            x0 = ((sample - np.sqrt(b_t) * mo) / np.sqrt(a_t)).astype(np.float32)
        elif self.prediction_type == "sample":
            x0 = mo.copy()
        else:  # v_prediction :213-216 — note model_output is overwritten FIRST, then reused for x0
            mo = (np.sqrt(a_t) * mo + np.sqrt(b_t) * sample).astype(np.float32)
            x0 = (np.sqrt(a_t) * sample - np.sqrt(b_t) * mo).astype(np.float32)
        if self.clip_sample:
            x0 = np.clip(x0, -1.0, 1.0).astype(np.float32)
        std_dev_t = f32(0.0)
        dirc = (np.sqrt(f32(f32(f32(1.0) - a_prev) - f32(std_dev_t * std_dev_t))) * mo).astype(np.float32)
        self.last_step = dict(pred_original_sample=x0, derivative=None)      # :246-250
        return (np.sqrt(a_prev) * x0 + dirc).astype(np.float32)

    def add_noise(self, x0, noise, t):
        ac = self.alphas_cumprod[int(t)]
        return (np.sqrt(ac) * x0 + np.sqrt(f32(f32(1.0) - ac)) * noise).astype(np.float32)


# --------------------------------------------------------------------------- DDPM


class DDPM(_Base):
    """schedulers/ddpm.rs.  REFERENCE_QUIRK: ``step`` indexes alphas/betas by the *position* of t
    in ``timesteps`` (:189-198), not by t."""

    kind = "ddpm"

    def __init__(self, T=1000, beta_start=0.00085, beta_end=0.012, schedule="scaled_linear",
                 prediction_type="epsilon", clip_sample=False, variance_type="fixed_small"):
        _check_ctor(T, beta_start, beta_end)
        self.T = T
        self.prediction_type = prediction_type
        self.clip_sample, self.variance_type = clip_sample, variance_type
        self.betas = make_betas(T, beta_start, beta_end, schedule)
        self.alphas = (f32(1.0) - self.betas).astype(np.float32)
        self.alphas_cumprod = alphas_cumprod_of(self.betas)
        self.timesteps = nd_linspace(0.0, T - 1, T)[::-1].copy()
        self._init_noise_sigma = f32(1.0)

    def needs_noise(self):
        return True

    def set_timesteps(self, n: int):
        """:177-186"""
        n = min(n, self.T)
        self.timesteps = nd_range(0.0, float(self.T), float(self.T // n))[::-1].copy()

    def scale_model_input(self, sample, t):
        return sample.astype(np.float32).copy()

    def get_variance(self, pos: int) -> np.float32:
        """:150-163"""
        a_t = self.alphas_cumprod[pos]
        a_prev = self.alphas_cumprod[pos - 1] if pos > 0 else f32(1.0)
        var = f32(f32(f32(f32(1.0) - a_prev) / f32(f32(1.0) - a_t)) * self.betas[pos])
        vt = self.variance_type
        if vt == "fixed_small":
            return f32(max(var, f32(1e-20)))
        if vt == "fixed_small_log":
            return f32(np.log(f32(max(var, f32(1e-20)))))
        if vt == "fixed_large":
            return self.betas[pos]
        if vt == "fixed_large_log":
            return f32(np.log(self.betas[pos]))
        return var

    def step(self, model_output, t, sample, noise=None):
        """:188-239"""
        pos = _position(self.timesteps, f32(t))
        a_t = self.alphas_cumprod[pos]
        a_prev = self.alphas_cumprod[pos - 1] if pos > 0 else f32(1.0)
        b_t, b_prev = f32(f32(1.0) - a_t), f32(f32(1.0) - a_prev)
        if self.prediction_type == "epsilon":
            x0 = ((sample - np.sqrt(b_t) * model_output) / np.sqrt(a_t)).astype(np.float32)
        elif self.prediction_type == "sample":
            x0 = model_output.astype(np.float32).copy()
        else:
            raise NotImplementedError("ddpm.rs:206 unimplemented!()")
        if self.clip_sample:
            x0 = np.clip(x0, -1.0, 1.0).astype(np.float32)
        c0 = f32(f32(np.sqrt(a_prev) * self.betas[pos]) / b_t)
        c1 = f32(f32(np.sqrt(self.alphas[pos]) * b_prev) / b_t)
        prev = (c0 * x0 + c1 * sample).astype(np.float32)
        self.last_step = dict(pred_original_sample=x0, derivative=None)      # :234-238
        if pos > 0:
            v = self.get_variance(pos)
            if self.variance_type != "fixed_small_log":
                v = f32(np.sqrt(v))
            prev = (prev + v * noise.astype(np.float32)).astype(np.float32)
        return prev

    def add_noise(self, x0, noise, t):
        pos = _position(self.timesteps, f32(t))
        ac = self.alphas_cumprod[pos]
        return (np.sqrt(ac) * x0 + np.sqrt(f32(f32(1.0) - ac)) * noise).astype(np.float32)


# --------------------------------------------------------------------------- LMS (absent from the reference)


class LMSDiscrete(EulerDiscrete):
    """K-LMS (Katherine Crowson's k-diffusion ``sample_lms`` / HF ``LMSDiscreteScheduler``).

    **Not in the reference** (README.md:35 advertises it; no file exists — SURVEY.md §0), so this
    is parity-unpinned against pyke by construction.  Built on the reference's Euler scaffolding
    (same constructor, sigma table, ``linspace(999,0,n)`` timesteps, input scaling).
    Coefficients: exact integration of the Lagrange basis polynomials (float64 here).
    """

    kind = "lms"
    ORDER = 4

    def set_timesteps(self, n: int):
        super().set_timesteps(n)
        self.derivatives: List[np.ndarray] = []

    def lms_coeffs(self, i: int) -> List[np.float32]:
        order = min(i + 1, self.ORDER)
        sig = self.sigmas.astype(np.float64)
        out = []
        for k in range(order):
            poly = np.poly1d([1.0])
            for j in range(order):
                if j == k:
                    continue
                poly = poly * np.poly1d([1.0, -sig[i - j]]) / (sig[i - k] - sig[i - j])
            P = poly.integ()
            out.append(f32(P(sig[i + 1]) - P(sig[i])))
        return out

    def step(self, model_output, t, sample, noise=None):
        assert self.scale_called
        i = _position(self.timesteps, f32(t))
        sigma = self.sigmas[i]
        pred = (sample - sigma * model_output).astype(np.float32)
        deriv = ((sample - pred) / sigma).astype(np.float32)
        self.last_step = dict(pred_original_sample=pred, derivative=deriv)
        self.derivatives.append(deriv)
        if len(self.derivatives) > self.ORDER:
            self.derivatives.pop(0)
        coeffs = self.lms_coeffs(i)
        prev = sample.astype(np.float32).copy()
        for c, d in zip(coeffs, reversed(self.derivatives)):
            prev = (prev + c * d).astype(np.float32)
        return prev


KINDS = {
    "euler": EulerDiscrete,
    "euler_a": EulerAncestralDiscrete,
    "dpmpp": DPMSolverMultistep,
    "ddim": DDIM,
    "ddpm": DDPM,
    "lms": LMSDiscrete,
}


def make(kind: str, **kw):
    """SD-v1 "optimized defaults" (euler_discrete.rs:261, dpm_solver_multistep.rs:404-419,
    ddim.rs:275-286, ddpm.rs:269) unless overridden."""
    return KINDS[kind](**kw)
