"""Host mirror of the reference's scheduler API (trait ``DiffusionScheduler``,
/root/reference/src/schedulers/mod.rs:154-192) over the C ABI.  Same names, argument meaning and
error behaviour (the reference panics on misuse; here ``SdeError`` is raised).

Table construction runs in the native library's host code; every tensor method
(``scale_model_input`` / ``step`` / ``add_noise``) runs the same HIP kernels as the fused denoise loop.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib

EULER, EULER_A, DPMPP, DDIM, DDPM, LMS = range(6)
BETA = {"linear": 0, "scaled_linear": 1, "squaredcos_cap_v2": 2}
PRED = {"epsilon": 0, "sample": 1, "v_prediction": 2}
VARIANCE = {"fixed_small": 0, "fixed_small_log": 1, "fixed_large": 2, "fixed_large_log": 3, "learned": 4}
DPM_ALGORITHM = {"dpmsolver++": 0, "dpmsolver": 1}   # DPMSolverAlgorithmType (dpm_solver_multistep.rs:30-36)
DPM_SOLVER = {"midpoint": 0, "heun": 1}              # DPMSolverType (:43-49)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


class SchedulerStepOutput:
    """schedulers/mod.rs:109-150 (accessor names kept; ``prev_sample_mean`` is None in every scheduler of the reference)."""

    def __init__(self, prev_sample, pred_original_sample=None, derivative=None):
        self._prev, self._x0, self._d = prev_sample, pred_original_sample, derivative

    def prev_sample(self):
        return self._prev

    def pred_original_sample(self):
        return self._x0

    def prev_sample_mean(self):
        return None

    def derivative(self):
        return self._d


class DiffusionScheduler:
    KIND = -1

    def __init__(self, num_train_timesteps: Optional[int] = None, beta_start: Optional[float] = None,
                 beta_end: Optional[float] = None, beta_schedule: Optional[str] = None,
                 prediction_type: Optional[str] = None, sd_v1: bool = False, **config):
        L = _lib.lib()
        cfg = _lib.SchedCfg()
        (L.sde_sched_cfg_sd_v1 if sd_v1 else L.sde_sched_cfg_default)(self.KIND, C.byref(cfg))
        if num_train_timesteps is not None:
            cfg.num_train_timesteps = num_train_timesteps
        if beta_start is not None:
            cfg.beta_start = beta_start
        if beta_end is not None:
            cfg.beta_end = beta_end
        if beta_schedule is not None:
            cfg.beta_schedule = BETA[beta_schedule]
        if prediction_type is not None:
            cfg.prediction_type = PRED[prediction_type]
        for k, v in config.items():
            if k == "variance_type":
                v = VARIANCE[v]
            elif k == "algorithm_type" and isinstance(v, str):
                v = DPM_ALGORITHM[v]
            elif k == "solver_type" and isinstance(v, str):
                v = DPM_SOLVER[v]
            if not hasattr(cfg, k):
                raise TypeError(f"unknown scheduler option {k}")
            setattr(cfg, k, int(v))
        self.cfg = cfg
        h = C.c_void_p()
        _lib.check(L.sde_scheduler_create(self.KIND, C.byref(cfg), C.byref(h)))
        self._h = h

    @classmethod
    def stable_diffusion_v1_optimized_default(cls, **config):
        """SchedulerOptimizedDefaults (schedulers/mod.rs:195-200)."""
        return cls(sd_v1=True, **config)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().sde_scheduler_destroy(h)
            except Exception:   # interpreter shutdown: the ctypes module may already be torn down
                pass
            self._h = None

    # --- trait methods
    @staticmethod
    def order() -> int:
        return 1

    def set_timesteps(self, num_inference_steps: int):
        _lib.check(_lib.lib().sde_scheduler_set_timesteps(self._h, int(num_inference_steps)))

    def timesteps(self) -> np.ndarray:
        L = _lib.lib()
        n = L.sde_scheduler_num_timesteps(self._h)
        out = np.empty(n, np.float32)
        _lib.check(L.sde_scheduler_timesteps(self._h, out.ctypes.data, n))
        return out

    def sigmas(self) -> np.ndarray:
        L = _lib.lib()
        n = C.c_int()
        _lib.check(L.sde_scheduler_sigmas(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.float32)
        if n.value:
            _lib.check(L.sde_scheduler_sigmas(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def init_noise_sigma(self) -> float:
        return float(_lib.lib().sde_scheduler_init_noise_sigma(self._h))

    def __len__(self) -> int:
        return _lib.lib().sde_scheduler_len(self._h)

    def scale_model_input(self, sample, timestep) -> np.ndarray:
        x = _f32(sample)
        out = np.empty_like(x)
        _lib.check(_lib.lib().sde_scheduler_scale_model_input(self._h, x.ctypes.data, x.size, float(timestep), out.ctypes.data))
        return out

    def step(self, model_output, timestep, sample, noise=None) -> np.ndarray:
        """``rng`` of the reference becomes an explicit N(0,1) ``noise`` tensor: draw it with ``sde_rng_standard_normal``
        (the restated StdRng / StandardNormal stream of the reference) or inject any other draws."""
        mo, s = _f32(model_output), _f32(sample)
        nz = _f32(noise) if noise is not None else None
        out = np.empty_like(s)
        _lib.check(_lib.lib().sde_scheduler_step(self._h, mo.ctypes.data, float(timestep), s.ctypes.data,
                                                 nz.ctypes.data if nz is not None else None, s.size, out.ctypes.data))
        return out

    def step_output(self, model_output, timestep, sample, noise=None) -> "SchedulerStepOutput":
        """``step`` returning the whole ``SchedulerStepOutput`` (schedulers/mod.rs:109-150): ``pred_original_sample`` is None where
        the reference's step leaves it None (DPM-Solver); ``derivative`` (the Euler family's) is offered although the reference
        never fills that field."""
        mo, s = _f32(model_output), _f32(sample)
        nz = _f32(noise) if noise is not None else None
        prev, x0, d = np.empty_like(s), np.empty_like(s), np.empty_like(s)
        o = _lib.StepOutput(prev.ctypes.data, x0.ctypes.data, d.ctypes.data, 0, 0)
        _lib.check(_lib.lib().sde_scheduler_step_output(self._h, mo.ctypes.data, float(timestep), s.ctypes.data,
                                                        nz.ctypes.data if nz is not None else None, s.size, C.byref(o)))
        return SchedulerStepOutput(prev, x0 if o.has_pred_original_sample else None, d if o.has_derivative else None)

    def add_noise(self, original_samples, noise, timestep) -> np.ndarray:
        x, nz = _f32(original_samples), _f32(noise)
        out = np.empty_like(x)
        _lib.check(_lib.lib().sde_scheduler_add_noise(self._h, x.ctypes.data, nz.ctypes.data, x.size, float(timestep), out.ctypes.data))
        return out


class EulerDiscreteScheduler(DiffusionScheduler):
    KIND = EULER


class EulerAncestralDiscreteScheduler(DiffusionScheduler):
    KIND = EULER_A


class DPMSolverMultistepScheduler(DiffusionScheduler):
    KIND = DPMPP


class DDIMScheduler(DiffusionScheduler):
    KIND = DDIM


class DDPMScheduler(DiffusionScheduler):
    KIND = DDPM


class LMSDiscreteScheduler(DiffusionScheduler):
    """Advertised by the reference's README.md:35 but absent from its source; k-diffusion definition."""
    KIND = LMS


BY_NAME = {"euler": EulerDiscreteScheduler, "euler_a": EulerAncestralDiscreteScheduler, "dpmpp": DPMSolverMultistepScheduler,
           "ddim": DDIMScheduler, "ddpm": DDPMScheduler, "lms": LMSDiscreteScheduler}
