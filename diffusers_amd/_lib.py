"""ctypes binding of libsdengine.so (C ABI declared in include/sdengine.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C diffusers_amd/csrc``.
There is no fallback: if the shared object is missing, importing the compute API raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDE_LIB_PATH") or os.path.join(_HERE, "libsdengine.so")   # SDE_LIB_PATH: tuning hook (A/B builds)

c_float_p = C.POINTER(C.c_float)


class EngineOpts(C.Structure):
    _fields_ = [("device_id", C.c_int), ("dtype", C.c_int), ("force_naive", C.c_int), ("load_unet", C.c_int),
                ("load_vae", C.c_int), ("num_heads_override", C.c_int)]


class ModelInfo(C.Structure):
    _fields_ = [("unet_loaded", C.c_int), ("vae_loaded", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
                ("n_levels", C.c_int), ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int),
                ("use_linear_projection", C.c_int), ("block_out_channels", C.c_int * 8), ("num_heads", C.c_int * 8),
                ("down_attn", C.c_int * 8), ("vae_n_levels", C.c_int), ("vae_layers_per_block", C.c_int),
                ("vae_latent_channels", C.c_int), ("vae_out_channels", C.c_int), ("vae_block_out_channels", C.c_int * 8),
                ("unet_params", C.c_int64), ("vae_params", C.c_int64)]


class SchedCfg(C.Structure):
    _fields_ = [("num_train_timesteps", C.c_int), ("beta_start", C.c_float), ("beta_end", C.c_float),
                ("beta_schedule", C.c_int), ("prediction_type", C.c_int), ("reference_quirks", C.c_int),
                ("solver_order", C.c_int), ("lower_order_final", C.c_int), ("clip_sample", C.c_int),
                ("set_alpha_to_one", C.c_int), ("steps_offset", C.c_int), ("variance_type", C.c_int),
                ("algorithm_type", C.c_int), ("solver_type", C.c_int)]


CALLBACK = C.CFUNCTYPE(C.c_int, C.c_int, C.c_float, c_float_p, C.c_size_t, C.c_void_p)
EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)   # sde_exchange_fn
EXCHANGE_STREAM = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)   # sde_exchange_stream_fn


class Txt2ImgOpts(C.Structure):
    _fields_ = [("height", C.c_int), ("width", C.c_int), ("guidance_scale", C.c_float), ("has_rescale_cfg", C.c_int),
                ("rescale_cfg", C.c_float), ("steps", C.c_int), ("has_seed", C.c_int), ("seed", C.c_uint64),
                ("batch_size", C.c_int), ("ehs", C.c_void_p), ("S", C.c_int), ("D", C.c_int),
                ("init_latents", C.c_void_p), ("step_noise", C.c_void_p), ("buffers_on_device", C.c_int),
                ("skip_decode", C.c_int), ("callback_kind", C.c_int), ("callback_frequency", C.c_int),
                ("callback", CALLBACK), ("callback_user", C.c_void_p), ("noise_pred_out", C.c_void_p),
                ("latents_out", C.c_void_p), ("final_latents_out", C.c_void_p),
                ("cfg_half", C.c_int), ("exchange", EXCHANGE), ("exchange_user", C.c_void_p),
                ("trace_rows", C.c_int), ("step_noise_rows", C.c_int), ("start_step", C.c_int),
                ("exchange_stream", EXCHANGE_STREAM)]


class StepOutput(C.Structure):
    _fields_ = [("prev_sample", C.c_void_p), ("pred_original_sample", C.c_void_p), ("derivative", C.c_void_p),
                ("has_pred_original_sample", C.c_int), ("has_derivative", C.c_int)]


class RunStats(C.Structure):
    _fields_ = [("unet_calls", C.c_int), ("total_ms", C.c_float), ("unet_ms", C.c_float), ("vae_ms", C.c_float),
                ("host_enqueue_ms", C.c_float), ("unet_graph_launches", C.c_int)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# every symbol include/sdengine.h declares: name -> (restype, argtypes)
_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
SYMBOLS = {
    "sde_last_error": (C.c_char_p, []),
    "sde_version": (C.c_char_p, []),
    "sde_device_count": (_i, []),
    "sde_engine_opts_default": (None, [C.POINTER(EngineOpts)]),
    "sde_engine_create": (_i, [C.c_char_p, C.POINTER(EngineOpts), C.POINTER(_vp)]),
    "sde_engine_create_from_files": (_i, [C.c_char_p, C.c_char_p, C.POINTER(EngineOpts), C.POINTER(_vp)]),
    "sde_engine_destroy": (None, [_vp]),
    "sde_engine_info": (_i, [_vp, C.POINTER(ModelInfo)]),
    "sde_unet_forward": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _i, _i, _vp]),
    "sde_unet_forward_dev": (_i, [_vp, _vp, _i, _i, _i, _f, _vp, _i, _i, _vp]),
    "sde_vae_decode": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sde_vae_decode_dev": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sde_vae_decoder_run": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sde_engine_load_vae_encoder": (_i, [_vp, C.c_char_p]),
    "sde_vae_encoder_run": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sde_vae_encoder_out_shape": (_i, [_vp, _i, _i, _vp, _vp, _vp]),
    "sde_approximate_decode": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sde_approximate_decode_noalloc": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sde_sched_cfg_sd_v1": (None, [_i, C.POINTER(SchedCfg)]),
    "sde_sched_cfg_default": (None, [_i, C.POINTER(SchedCfg)]),
    "sde_scheduler_create": (_i, [_i, C.POINTER(SchedCfg), C.POINTER(_vp)]),
    "sde_scheduler_destroy": (None, [_vp]),
    "sde_scheduler_order": (_i, [_vp]),
    "sde_scheduler_len": (_i, [_vp]),
    "sde_scheduler_set_timesteps": (_i, [_vp, _i]),
    "sde_scheduler_num_timesteps": (_i, [_vp]),
    "sde_scheduler_timesteps": (_i, [_vp, _vp, _i]),
    "sde_scheduler_sigmas": (_i, [_vp, _vp, _i, C.POINTER(_i)]),
    "sde_scheduler_init_noise_sigma": (_f, [_vp]),
    "sde_scheduler_scale_model_input": (_i, [_vp, _vp, C.c_size_t, _f, _vp]),
    "sde_scheduler_step": (_i, [_vp, _vp, _f, _vp, _vp, C.c_size_t, _vp]),
    "sde_scheduler_add_noise": (_i, [_vp, _vp, _vp, C.c_size_t, _f, _vp]),
    "sde_scheduler_step_output": (_i, [_vp, _vp, _f, _vp, _vp, C.c_size_t, C.POINTER(StepOutput)]),
    "sde_txt2img_opts_default": (None, [C.POINTER(Txt2ImgOpts)]),
    "sde_txt2img_run": (_i, [_vp, _vp, C.POINTER(Txt2ImgOpts), _vp, C.POINTER(RunStats)]),
    "sde_rng_standard_normal": (_i, [C.c_uint64, C.c_uint64, _vp, C.c_size_t]),
    "sde_rng_chacha_words": (_i, [_vp, _i, _vp, C.c_size_t]),
    "sde_rng_seed_key": (_i, [C.c_uint64, _vp]),
    "sde_profile_enable": (_i, [_vp, _i]),
    "sde_profile_reset": (_i, [_vp]),
    "sde_profile_get": (_i, [_vp, C.POINTER(KernelStat), _i, C.POINTER(_i)]),
    "sde_synchronize": (_i, [_vp]),
    "sde_test_conv": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp, C.c_size_t]),
    "sde_test_conv_res": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _vp, C.c_size_t]),
    "sde_test_geglu": (_i, [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp, C.c_size_t]),
    "sde_test_attention": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.c_size_t]),
}

_lib = None


class SdeError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"[sde_status {status}] {message}")
        self.status = status
        self.message = message


def lib() -> C.CDLL:
    """Loads the in-tree shared library; raises (never falls back) if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"or `make -C diffusers_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int):
    if status != 0:
        raise SdeError(status, lib().sde_last_error().decode(errors="replace"))
