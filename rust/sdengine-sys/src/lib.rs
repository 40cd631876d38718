//! FFI declarations of `include/sdengine.h` — UNCOMPILED (no Rust toolchain in the build image).
//! One `#[repr(C)]` struct per C struct (same field order), one `extern "C"` item per entry point.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_double, c_float, c_int, c_void};

#[repr(C)] pub struct sde_engine { _p: [u8; 0] }
#[repr(C)] pub struct sde_scheduler { _p: [u8; 0] }

// sde_status
pub const SDE_OK: c_int = 0;
pub const SDE_ERR_INVALID_ARGUMENT: c_int = 1;
pub const SDE_ERR_IO: c_int = 2;
pub const SDE_ERR_CONFIG: c_int = 3;
pub const SDE_ERR_MODEL: c_int = 4;
pub const SDE_ERR_DEVICE: c_int = 5;
pub const SDE_ERR_SCHEDULER: c_int = 6;
pub const SDE_ERR_CANCELLED: c_int = 7;
pub const SDE_ERR_EXCHANGE: c_int = 8;
// sde_dtype / sde_scheduler_kind / sde_beta_schedule / sde_prediction_type / sde_callback_kind / DPM-Solver enums
pub const SDE_F16: c_int = 0; pub const SDE_BF16: c_int = 1; pub const SDE_F32: c_int = 2;
pub const SDE_SCHED_EULER: c_int = 0; pub const SDE_SCHED_EULER_A: c_int = 1; pub const SDE_SCHED_DPMPP: c_int = 2;
pub const SDE_SCHED_DDIM: c_int = 3; pub const SDE_SCHED_DDPM: c_int = 4; pub const SDE_SCHED_LMS: c_int = 5;
pub const SDE_BETA_LINEAR: c_int = 0; pub const SDE_BETA_SCALED_LINEAR: c_int = 1; pub const SDE_BETA_SQUAREDCOS_CAP_V2: c_int = 2;
pub const SDE_PRED_EPSILON: c_int = 0; pub const SDE_PRED_SAMPLE: c_int = 1; pub const SDE_PRED_V: c_int = 2;
pub const SDE_CB_NONE: c_int = 0; pub const SDE_CB_PROGRESS: c_int = 1; pub const SDE_CB_LATENTS: c_int = 2;
pub const SDE_CB_DECODED: c_int = 3; pub const SDE_CB_APPROXIMATE: c_int = 4;
pub const SDE_DPM_SOLVER_PP: c_int = 0; pub const SDE_DPM_SOLVER: c_int = 1;
pub const SDE_DPM_MIDPOINT: c_int = 0; pub const SDE_DPM_HEUN: c_int = 1;
pub const SDE_NOALLOC_SCRATCH_BYTES: usize = 8 << 20;

#[repr(C)] #[derive(Clone, Copy)]
pub struct sde_engine_opts { pub device_id: c_int, pub dtype: c_int, pub force_naive: c_int,
                             pub load_unet: c_int, pub load_vae: c_int, pub num_heads_override: c_int }

#[repr(C)] #[derive(Clone, Copy)]
pub struct sde_model_info { pub unet_loaded: c_int, pub vae_loaded: c_int, pub in_channels: c_int, pub out_channels: c_int,
    pub n_levels: c_int, pub layers_per_block: c_int, pub cross_attention_dim: c_int, pub use_linear_projection: c_int,
    pub block_out_channels: [c_int; 8], pub num_heads: [c_int; 8], pub down_attn: [c_int; 8],
    pub vae_n_levels: c_int, pub vae_layers_per_block: c_int, pub vae_latent_channels: c_int, pub vae_out_channels: c_int,
    pub vae_block_out_channels: [c_int; 8], pub unet_params: i64, pub vae_params: i64 }

#[repr(C)] #[derive(Clone, Copy)]
pub struct sde_sched_cfg { pub num_train_timesteps: c_int, pub beta_start: c_float, pub beta_end: c_float,
    pub beta_schedule: c_int, pub prediction_type: c_int, pub reference_quirks: c_int, pub solver_order: c_int,
    pub lower_order_final: c_int, pub clip_sample: c_int, pub set_alpha_to_one: c_int, pub steps_offset: c_int,
    pub variance_type: c_int, pub algorithm_type: c_int, pub solver_type: c_int }

pub type sde_callback_fn = Option<unsafe extern "C" fn(c_int, c_float, *const c_float, usize, *mut c_void) -> c_int>;
pub type sde_exchange_fn = Option<unsafe extern "C" fn(*mut c_float, usize, c_int, *mut c_void) -> c_int>;
/// stream-ordered form: (d_model_out, half_elems, my_half, hipStream_t, user)
pub type sde_exchange_stream_fn = Option<unsafe extern "C" fn(*mut c_float, usize, c_int, *mut c_void, *mut c_void) -> c_int>;

#[repr(C)]
pub struct sde_txt2img_opts { pub height: c_int, pub width: c_int, pub guidance_scale: c_float,
    pub has_rescale_cfg: c_int, pub rescale_cfg: c_float, pub steps: c_int, pub has_seed: c_int, pub seed: u64,
    pub batch_size: c_int, pub ehs: *const c_float, pub s: c_int, pub d: c_int,
    pub init_latents: *const c_float, pub step_noise: *const c_float, pub buffers_on_device: c_int,
    pub skip_decode: c_int, pub callback_kind: c_int, pub callback_frequency: c_int,
    pub callback: sde_callback_fn, pub callback_user: *mut c_void,
    pub noise_pred_out: *mut c_float, pub latents_out: *mut c_float, pub final_latents_out: *mut c_float,
    pub cfg_half: c_int, pub exchange: sde_exchange_fn, pub exchange_user: *mut c_void,
    pub trace_rows: c_int, pub step_noise_rows: c_int,
    /// image-to-image: run timesteps[start_step..] from already-noised `init_latents` (no counterpart in the reference's run)
    pub start_step: c_int, pub exchange_stream: sde_exchange_stream_fn }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct sde_run_stats { pub unet_calls: c_int, pub total_ms: c_float, pub unet_ms: c_float, pub vae_ms: c_float, pub host_enqueue_ms: c_float, pub unet_graph_launches: c_int }

#[repr(C)] #[derive(Clone, Copy)]
pub struct sde_step_output { pub prev_sample: *mut c_float, pub pred_original_sample: *mut c_float, pub derivative: *mut c_float, pub has_pred_original_sample: c_int, pub has_derivative: c_int }

#[repr(C)] #[derive(Clone, Copy)]
pub struct sde_kernel_stat { pub name: [c_char; 48], pub launches: u64, pub total_ms: c_double, pub flops: c_double, pub bytes: c_double }

extern "C" {
    pub fn sde_last_error() -> *const c_char;
    pub fn sde_version() -> *const c_char;
    pub fn sde_device_count() -> c_int;
    pub fn sde_engine_opts_default(o: *mut sde_engine_opts);
    pub fn sde_engine_create(model_dir: *const c_char, o: *const sde_engine_opts, out: *mut *mut sde_engine) -> c_int;
    pub fn sde_engine_create_from_files(unet: *const c_char, vae: *const c_char, o: *const sde_engine_opts, out: *mut *mut sde_engine) -> c_int;
    pub fn sde_engine_destroy(e: *mut sde_engine);
    pub fn sde_engine_info(e: *const sde_engine, info: *mut sde_model_info) -> c_int;
    pub fn sde_unet_forward(e: *mut sde_engine, sample: *const c_float, n: c_int, h: c_int, w: c_int, timestep: c_float,
                            ehs: *const c_float, s: c_int, d: c_int, out: *mut c_float, out_capacity: usize) -> c_int;
    pub fn sde_unet_forward_dev(e: *mut sde_engine, d_sample: *const c_float, n: c_int, h: c_int, w: c_int, timestep: c_float,
                                d_ehs: *const c_float, s: c_int, d: c_int, d_out: *mut c_float) -> c_int;
    pub fn sde_vae_decode(e: *mut sde_engine, latents: *const c_float, b: c_int, h: c_int, w: c_int, images: *mut c_float) -> c_int;
    pub fn sde_vae_decode_dev(e: *mut sde_engine, d_latents: *const c_float, b: c_int, h: c_int, w: c_int, d_images: *mut c_float) -> c_int;
    pub fn sde_vae_decoder_run(e: *mut sde_engine, z: *const c_float, b: c_int, h: c_int, w: c_int, sample: *mut c_float) -> c_int;
    /// `vae_encoder: Option<Session>` (impl_main.rs:58,140-150); NULL path = the model dir's `[vae] encoder`
    pub fn sde_engine_load_vae_encoder(e: *mut sde_engine, onnx_path: *const c_char) -> c_int;
    /// `vae_encoder.run([sample]) -> latent_sample`; the graph's RandomNormalLike is the `noise` argument (NULL = posterior mean)
    pub fn sde_vae_encoder_run(e: *mut sde_engine, sample: *const c_float, b: c_int, h: c_int, w: c_int, noise: *const c_float, latent_sample: *mut c_float) -> c_int;
    pub fn sde_vae_encoder_out_shape(e: *const sde_engine, h_img: c_int, w_img: c_int, latent_channels: *mut c_int, h: *mut c_int, w: *mut c_int) -> c_int;
    pub fn sde_approximate_decode(e: *mut sde_engine, latents: *const c_float, b: c_int, h: c_int, w: c_int, images: *mut c_float) -> c_int;
    pub fn sde_approximate_decode_noalloc(e: *mut sde_engine, latents: *const c_float, b: c_int, h: c_int, w: c_int, images: *mut c_float) -> c_int;
    pub fn sde_sched_cfg_sd_v1(kind: c_int, cfg: *mut sde_sched_cfg);
    pub fn sde_sched_cfg_default(kind: c_int, cfg: *mut sde_sched_cfg);
    pub fn sde_scheduler_create(kind: c_int, cfg: *const sde_sched_cfg, out: *mut *mut sde_scheduler) -> c_int;
    pub fn sde_scheduler_destroy(s: *mut sde_scheduler);
    pub fn sde_scheduler_order(s: *const sde_scheduler) -> c_int;
    pub fn sde_scheduler_len(s: *const sde_scheduler) -> c_int;
    pub fn sde_scheduler_set_timesteps(s: *mut sde_scheduler, n: c_int) -> c_int;
    pub fn sde_scheduler_num_timesteps(s: *const sde_scheduler) -> c_int;
    pub fn sde_scheduler_timesteps(s: *const sde_scheduler, out: *mut c_float, capacity: c_int) -> c_int;
    pub fn sde_scheduler_sigmas(s: *const sde_scheduler, out: *mut c_float, capacity: c_int, n_out: *mut c_int) -> c_int;
    pub fn sde_scheduler_init_noise_sigma(s: *const sde_scheduler) -> c_float;
    pub fn sde_scheduler_scale_model_input(s: *mut sde_scheduler, sample: *const c_float, n: usize, timestep: c_float, out: *mut c_float) -> c_int;
    pub fn sde_scheduler_step(s: *mut sde_scheduler, model_output: *const c_float, timestep: c_float, sample: *const c_float,
                              noise: *const c_float, n: usize, prev_sample: *mut c_float) -> c_int;
    pub fn sde_scheduler_add_noise(s: *mut sde_scheduler, original: *const c_float, noise: *const c_float, n: usize,
                                   timestep: c_float, out: *mut c_float) -> c_int;
    pub fn sde_scheduler_step_output(s: *mut sde_scheduler, model_output: *const c_float, timestep: c_float, sample: *const c_float,
                                     noise: *const c_float, n: usize, out: *mut sde_step_output) -> c_int;
    pub fn sde_txt2img_opts_default(o: *mut sde_txt2img_opts);
    pub fn sde_txt2img_run(e: *mut sde_engine, s: *mut sde_scheduler, o: *const sde_txt2img_opts,
                           images: *mut c_float, stats: *mut sde_run_stats) -> c_int;
    pub fn sde_rng_standard_normal(seed: u64, skip: u64, out: *mut c_float, n: usize) -> c_int;
    pub fn sde_rng_chacha_words(key: *const u8, rounds: c_int, out: *mut u32, n_words: usize) -> c_int;
    pub fn sde_rng_seed_key(seed: u64, key_out: *mut u8) -> c_int;
    pub fn sde_profile_enable(e: *mut sde_engine, enable: c_int) -> c_int;
    pub fn sde_profile_reset(e: *mut sde_engine) -> c_int;
    pub fn sde_profile_get(e: *mut sde_engine, out: *mut sde_kernel_stat, capacity: c_int, n_out: *mut c_int) -> c_int;
    pub fn sde_synchronize(e: *mut sde_engine) -> c_int;
    pub fn sde_test_conv(e: *mut sde_engine, naive: c_int, x: *const c_float, nb: c_int, hi: c_int, wi: c_int, cin: c_int,
                         w: *const c_float, bias: *const c_float, cout: c_int, ksize: c_int, stride: c_int, upsample: c_int,
                         out: *mut c_float, out_capacity: usize) -> c_int;
    pub fn sde_test_conv_res(e: *mut sde_engine, naive: c_int, x: *const c_float, nb: c_int, hi: c_int, wi: c_int, cin: c_int,
                             w: *const c_float, bias: *const c_float, cout: c_int, ksize: c_int, residual: *const c_float, split: c_int,
                             out: *mut c_float, out_capacity: usize) -> c_int;
    pub fn sde_test_geglu(e: *mut sde_engine, naive: c_int, x: *const c_float, m: c_int, k: c_int, w: *const c_float, bias: *const c_float,
                          n: c_int, gelu_tanh: c_int, out: *mut c_float, out_capacity: usize) -> c_int;
    pub fn sde_test_attention(e: *mut sde_engine, naive: c_int, q: *const c_float, k: *const c_float, v: *const c_float,
                              nb: c_int, heads: c_int, sq: c_int, skv: c_int, d: c_int, out: *mut c_float, out_capacity: usize) -> c_int;
}
