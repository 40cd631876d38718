/* sdengine.h — C ABI of the MI355X-native Stable Diffusion denoising engine.
 *
 * Drop-in boundary for the hot path of pyke Diffusers (reference paths are relative to the
 * reference repository root).  A Rust `-sys` crate (or any FFI) binds exactly these entry points;
 * INTEGRATION.md shows the reference-side stub.  Conventions:
 *   - plain C types only; every tensor argument is a caller-owned buffer of f32, row-major,
 *     NCHW unless stated otherwise — the same layout the reference hands to `ort::Session::run`;
 *   - `*_dev` variants take DEVICE pointers (hipMalloc'd / torch CUDA tensors) instead of host ones;
 *   - every function returns sde_status; on failure `sde_last_error()` (thread-local) explains;
 *     nothing aborts or throws across the boundary (the reference returns anyhow::Error or panics);
 *   - handles are opaque, owned by the library, released only by the matching *_destroy.
 */
#ifndef SDENGINE_H
#define SDENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sde_status {
    SDE_OK = 0,
    SDE_ERR_INVALID_ARGUMENT = 1, /* bad sizes / null pointers (reference: anyhow::bail, impl_txt2img.rs:251-253) */
    SDE_ERR_IO = 2,               /* model directory / file problems (reference: `?` on fs/ORT load, impl_main.rs:111,136-157) */
    SDE_ERR_CONFIG = 3,           /* pyke-diffusers.toml problems, "bad framework" (impl_main.rs:113-122) */
    SDE_ERR_MODEL = 4,            /* ONNX parse / weight-binding failure */
    SDE_ERR_DEVICE = 5,           /* HIP runtime error, no GPU, extension built for another arch */
    SDE_ERR_SCHEDULER = 6,        /* scheduler misuse that panics in the reference (euler_discrete.rs:140,181).  An EMPTY schedule (DDIM
                                   * with one inference step: ddim.rs:184 yields n - 1 timesteps) is NOT an error in parity mode
                                   * (sde_sched_cfg.reference_quirks = 1, the default): as in the reference no step runs and
                                   * init_noise_sigma * latents are decoded; with reference_quirks = 0 it is refused with this status */
    SDE_ERR_CANCELLED = 7,        /* a callback returned 0 (reference: `break`, impl_txt2img.rs:319-321) */
    SDE_ERR_EXCHANGE = 8          /* the CFG-pair exchange hook reported a failure (no counterpart in the reference) */
} sde_status;

typedef struct sde_engine sde_engine;
typedef struct sde_scheduler sde_scheduler;

/* Library ---------------------------------------------------------------------------------- */
const char* sde_last_error(void);
const char* sde_version(void);
/* Number of visible HIP devices (0 when none / no driver).  Never fails. */
int sde_device_count(void);

/* Engine (replaces the `unet: Session` and `vae_decoder: Session` fields, impl_main.rs:59,64) - */
typedef enum sde_dtype { SDE_F16 = 0, SDE_BF16 = 1, SDE_F32 = 2 } sde_dtype;

typedef struct sde_engine_opts {
    int device_id;       /* HIP device ordinal (the new "native MI355X(device_id)" DiffusionDevice, lib.rs:81-107) */
    int dtype;           /* sde_dtype: activation/weight storage inside the engine; I/O is always f32.  SDE_F16 (default): MFMA kernels,
                          * fp32 accumulation, the benchmarked path.  SDE_F32: every tensor and every product in fp32 (v_mfma_f32_16x16x4_f32 tiles, f32path.hip), the
                          * precision of the reference's own graphs (sd2pyke.py:86,120-125); bar 1e-4.  SDE_BF16: not implemented (refused).
                          * Limit of SDE_F16: its 1e-2 bar assumes every GroupNorm group holds >= 8 values (channels per group x H x W at the
                          * deepest level).  SD v1.x / v2.x meet that from 16 x 16 pixel images on (40 channels per group at the deepest level);
                          * a model with one channel per group and a 1 x 1 deepest level does not (fp16 rounding noise is multiplied by
                          * rstd ~ eps^-1/2 = 316): a relative error of 0.5 was measured on such a draw, on which the SDE_F32 engine itself sits at
                          * 4e-4 (tests/test_gpu_fuzz.py::test_degenerate_groupnorm_draw_is_pinned) -- use SDE_F32 for such models. */
    int force_naive;     /* 1 = run every contraction on the plain (non-MFMA) kernels (debug / cross-check) */
    int load_unet;       /* 0 skips unet.onnx  (engine then only serves sde_vae_decode) */
    int load_vae;        /* 0 skips vae_decoder.onnx */
    int num_heads_override; /* >0 forces the attention head count when the graph does not reveal it */
} sde_engine_opts;

void sde_engine_opts_default(sde_engine_opts* opts);

/* StableDiffusionPipeline::new (impl_main.rs:109-181): parse <model_dir>/pyke-diffusers.toml
 * (config.rs:17-117), load [unet].path and [vae].decoder, upload weights. */
sde_status sde_engine_create(const char* model_dir, const sde_engine_opts* opts, sde_engine** out);
/* replace_unet / replace_vae (impl_main.rs:271-276,306-324): load single ONNX files. */
sde_status sde_engine_create_from_files(const char* unet_onnx, const char* vae_decoder_onnx,
                                        const sde_engine_opts* opts, sde_engine** out);
void sde_engine_destroy(sde_engine* e);

/* Introspection of what the loader inferred from the graphs. */
typedef struct sde_model_info {
    int unet_loaded, vae_loaded;
    int in_channels, out_channels, n_levels, layers_per_block, cross_attention_dim, use_linear_projection;
    int block_out_channels[8];
    int num_heads[8];
    int down_attn[8];
    int vae_n_levels, vae_layers_per_block, vae_latent_channels, vae_out_channels;
    int vae_block_out_channels[8];
    int64_t unet_params, vae_params;
} sde_model_info;
sde_status sde_engine_info(const sde_engine* e, sde_model_info* info);

/* The ORT call `unet.run([sample, timestep, encoder_hidden_states]) -> out_sample`
 * (impl_txt2img.rs:283; I/O names and dtypes sd2pyke.py:113-125,294-295).
 *   sample f32 [N,4,h,w]; timestep: one f32, cast to int64 inside the graph (= truncation,
 *   sd2pyke.py:121); ehs f32 [N,S,D]; out f32 [N,4,h,w]. */
sde_status sde_unet_forward(sde_engine* e, const float* sample, int N, int h, int w, float timestep,
                            const float* ehs, int S, int D, float* out);
sde_status sde_unet_forward_dev(sde_engine* e, const float* d_sample, int N, int h, int w, float timestep,
                                const float* d_ehs, int S, int D, float* d_out);

/* All of `decode_latents` (impl_main.rs:398-413) + `to_image` (:377-382): latents f32 [B,4,h,w]
 * -> *1/0.18215 -> VAE decoder -> NHWC, /2+0.5, clamp[0,1] -> images f32 [B,8h,8w,3]. */
sde_status sde_vae_decode(sde_engine* e, const float* latents, int B, int h, int w, float* images_nhwc);
sde_status sde_vae_decode_dev(sde_engine* e, const float* d_latents, int B, int h, int w, float* d_images_nhwc);
/* The bare ORT call `vae_decoder.run([latent_sample]) -> sample` (impl_main.rs:403): f32 [B,3,8h,8w], NCHW, no post-processing. */
sde_status sde_vae_decoder_run(sde_engine* e, const float* latent_sample, int B, int h, int w, float* sample_nchw);
/* `approximate_decode_latents` (impl_main.rs:385-395): 4->3 channel linear map + clamp -> f32 [B,h,w,3]. */
sde_status sde_approximate_decode(sde_engine* e, const float* latents, int B, int h, int w, float* images_nhwc);

/* The same as sde_approximate_decode for callers that cannot allocate device memory (tight serving loops, harnesses that forbid
 * hipMalloc, processes whose memory pool is fixed at start-up): it has NO counterpart in the reference.  It calls no hipMalloc /
 * hipFree: the two staging tensors live in a static device array of SDE_NOALLOC_SCRATCH_BYTES that every GPU the library runs on
 * holds until the process ends.  It is a BLOCKING host-pointer call like its sibling (engine lock, two asynchronous copies on the
 * engine's own stream, a stream synchronise), so it is not something to issue from inside a caller's stream capture.  B*h*w pixels need 256-aligned(16*B*h*w) + 12*B*h*w bytes; a larger request is
 * refused with SDE_ERR_INVALID_ARGUMENT (use sde_approximate_decode).  All calls on one GPU share that GPU's array: calls from
 * this process are serialised on it, calls that overlap in any other way (e.g. a copy of the library loaded twice) would race. */
#define SDE_NOALLOC_SCRATCH_BYTES (8u << 20)
sde_status sde_approximate_decode_noalloc(sde_engine* e, const float* latents, int B, int h, int w, float* images_nhwc);

/* VAE encoder: the reference opens this session when the model dir names one (`vae_encoder: Option<Session>`, impl_main.rs:58,140-150;
 * `replace_vae`, :315-320) and never runs it -- impl_img2img.rs only builds options.  Loading is therefore a call of its own.
 * onnx_path NULL: the `[vae] encoder` entry of the model dir the engine was created from (config.rs:91). */
sde_status sde_engine_load_vae_encoder(sde_engine* e, const char* onnx_path);
/* The bare session call of the exported graph (sd2pyke.py:181-187,325-346): `sample` f32 [B,3,H,W] in [-1,1] -> `latent_sample`
 * f32 [B,4,H/8,W/8] = mean + exp(0.5 clamp(logvar, -30, 20)) * noise.  The graph draws its noise with RandomNormalLike; here the
 * caller passes it (f32 [B,4,H/8,W/8], e.g. from sde_rng_standard_normal), or NULL for the posterior mean.  Not scaled by 0.18215. */
sde_status sde_vae_encoder_run(sde_engine* e, const float* sample_nchw, int B, int H, int W, const float* noise, float* latent_sample);
/* shape of latent_sample for an H x W image: every level but the last halves the size as a pad-(0,1,0,1), stride-2, 3x3 window does */
sde_status sde_vae_encoder_out_shape(const sde_engine* e, int H, int W, int* latent_channels, int* h, int* w);

/* Schedulers (trait DiffusionScheduler, schedulers/mod.rs:154-192) -------------------------- */
typedef enum sde_scheduler_kind {
    SDE_SCHED_EULER = 0,   /* euler_discrete.rs */
    SDE_SCHED_EULER_A = 1, /* euler_ancestral_discrete.rs */
    SDE_SCHED_DPMPP = 2,   /* dpm_solver_multistep.rs (DPM-Solver++ multistep) */
    SDE_SCHED_DDIM = 3,    /* ddim.rs */
    SDE_SCHED_DDPM = 4,    /* ddpm.rs */
    SDE_SCHED_LMS = 5      /* not in the reference (README.md:35 only); k-diffusion definition */
} sde_scheduler_kind;
typedef enum sde_beta_schedule { SDE_BETA_LINEAR = 0, SDE_BETA_SCALED_LINEAR = 1, SDE_BETA_SQUAREDCOS_CAP_V2 = 2 } sde_beta_schedule;
typedef enum sde_prediction_type { SDE_PRED_EPSILON = 0, SDE_PRED_SAMPLE = 1, SDE_PRED_V = 2 } sde_prediction_type;

typedef struct sde_sched_cfg {
    int num_train_timesteps;   /* 1000 */
    float beta_start, beta_end;
    int beta_schedule;         /* sde_beta_schedule */
    int prediction_type;       /* sde_prediction_type (ignored by Euler/Euler-a/LMS: epsilon only, as in the reference) */
    int reference_quirks;      /* 1 (default) = bug-for-bug with the reference; 0 = "fixed" variants */
    /* DPM-Solver++ */
    int solver_order;          /* 2 */
    int lower_order_final;     /* 1 */
    /* DDIM */
    int clip_sample, set_alpha_to_one, steps_offset;
    /* DDPM */
    int variance_type;         /* 0 fixed_small, 1 fixed_small_log, 2 fixed_large, 3 fixed_large_log, 4 learned */
    /* DPM-Solver, continued (dpm_solver_multistep.rs:30-49,71-74) */
    int algorithm_type;        /* sde_dpm_algorithm: 0 DPMSolverPlusPlus (default), 1 DPMSolver (:232-244,255,282-283,319-324) */
    int solver_type;           /* sde_dpm_solver_type: 0 Midpoint (default), 1 Heun (:279,283); second-order update only */
} sde_sched_cfg;
typedef enum sde_dpm_algorithm { SDE_DPM_SOLVER_PP = 0, SDE_DPM_SOLVER = 1 } sde_dpm_algorithm;
typedef enum sde_dpm_solver_type { SDE_DPM_MIDPOINT = 0, SDE_DPM_HEUN = 1 } sde_dpm_solver_type;

/* `stable_diffusion_v1_optimized_default()` of the given kind (euler_discrete.rs:261,
 * dpm_solver_multistep.rs:404-419, ddim.rs:275-286, ddpm.rs:269). */
void sde_sched_cfg_sd_v1(int kind, sde_sched_cfg* cfg);
/* `Default::default()` of the given kind (linear 0.0001..0.02; e.g. euler_discrete.rs:41-45). */
void sde_sched_cfg_default(int kind, sde_sched_cfg* cfg);

sde_status sde_scheduler_create(int kind, const sde_sched_cfg* cfg, sde_scheduler** out);
void sde_scheduler_destroy(sde_scheduler* s);
int sde_scheduler_order(const sde_scheduler* s);                       /* DiffusionScheduler::order */
int sde_scheduler_len(const sde_scheduler* s);                         /* ::len (num_train_timesteps) */
sde_status sde_scheduler_set_timesteps(sde_scheduler* s, int num_inference_steps);
int sde_scheduler_num_timesteps(const sde_scheduler* s);               /* timesteps().len() */
sde_status sde_scheduler_timesteps(const sde_scheduler* s, float* out, int capacity); /* as f32 (ToPrimitive::to_f32) */
sde_status sde_scheduler_sigmas(const sde_scheduler* s, float* out, int capacity, int* n_out); /* Euler family table */
float sde_scheduler_init_noise_sigma(const sde_scheduler* s);
/* Host f32 mirrors of the trait's tensor methods (n = number of elements), for callers that keep
 * their own loop and for parity tests.  `noise` may be NULL for deterministic schedulers. */
sde_status sde_scheduler_scale_model_input(sde_scheduler* s, const float* sample, size_t n, float timestep, float* out);
sde_status sde_scheduler_step(sde_scheduler* s, const float* model_output, float timestep, const float* sample,
                              const float* noise, size_t n, float* prev_sample);
sde_status sde_scheduler_add_noise(sde_scheduler* s, const float* original, const float* noise, size_t n,
                                   float timestep, float* out);
/* `SchedulerStepOutput` (schedulers/mod.rs:109-150): the same step with the struct's optional fields on request.  prev_sample is
 * required; pred_original_sample / derivative may be NULL (not wanted).  On return has_pred_original_sample says whether the
 * reference's step fills `pred_original_sample: Some(..)` -- Euler (euler_discrete.rs:216-220), Euler-a
 * (euler_ancestral_discrete.rs:206-210), DDIM (ddim.rs:246-250), DDPM (ddpm.rs:234-238) and LMS do, DPM-Solver does not
 * (dpm_solver_multistep.rs:379) -- and the buffer, if given, then holds it (after clipping, where clip_sample applies).
 * `derivative` is (sample - pred_original_sample) / sigma of the Euler family (euler_discrete.rs:210; LMS: the derivative it
 * stores): the reference computes it and leaves the struct's field None, here it is on offer (has_derivative = 1 for those kinds).
 * `prev_sample_mean` is None in every scheduler of the reference and has no counterpart. */
typedef struct sde_step_output {
    float* prev_sample;            /* f32 [n], required */
    float* pred_original_sample;   /* f32 [n] or NULL */
    float* derivative;             /* f32 [n] or NULL */
    int has_pred_original_sample;  /* out */
    int has_derivative;            /* out */
} sde_step_output;
sde_status sde_scheduler_step_output(sde_scheduler* s, const float* model_output, float timestep, const float* sample,
                                     const float* noise, size_t n, sde_step_output* out);

/* txt2img (StableDiffusionTxt2ImgOptions::run, impl_txt2img.rs:246-327) --------------------- */
typedef enum sde_callback_kind {  /* StableDiffusionCallback, pipelines/stable_diffusion/mod.rs:41-86 */
    SDE_CB_NONE = 0, SDE_CB_PROGRESS = 1, SDE_CB_LATENTS = 2, SDE_CB_DECODED = 3, SDE_CB_APPROXIMATE = 4
} sde_callback_kind;
/* step index, timestep, payload (NULL / latents [B,4,h,w] / images [B,H,W,3] / images [B,h,w,3]),
 * payload element count, user pointer.  Return 0 to stop the loop (reference: `false` -> break). */
typedef int (*sde_callback_fn)(int step, float timestep, const float* payload, size_t payload_len, void* user);
/* Callbacks (and the exchange hook below) run on the calling thread while the engine's lock is held, as the reference runs
 * them inside `run(&self, ..)`: a callback must NOT call back into the same engine (sde_unet_forward, sde_vae_decode,
 * sde_approximate_decode, sde_txt2img_run ...) -- such a call is refused with SDE_ERR_INVALID_ARGUMENT ("re-entered")
 * instead of deadlocking.  Use the Decoded / ApproximateDecoded kinds for previews, or a second engine. */

/* CFG-pair split (BASELINE config 3): two engines on two GPUs each run ONE half of the classifier-free-guidance
 * batch (impl_txt2img.rs:273-301 computes both in one session.run) and swap the halves once per step.  The engine calls this
 * hook on the calling thread after its UNet call, with its stream idle: `d_model_out` is the DEVICE buffer f32 [2][B,4,h,w]
 * (uncond half first); half `my_half` (0 or 1) is filled.  The hook fills the other half with the peer's output (e.g. a
 * 2-rank RCCL all-gather over xGMI) and returns non-zero once that transfer has completed; 0 aborts the run.
 * No counterpart in the reference (single ORT session). */
typedef int (*sde_exchange_fn)(float* d_model_out, size_t half_elems, int my_half, void* user);
/* The stream-ordered form of the same hook (preferred when the transport can be enqueued): the engine does NOT synchronise its
 * stream.  `hip_stream` is the engine's hipStream_t; the UNet call that fills half `my_half` has been enqueued on it, and the
 * fused CFG + scheduler kernel that reads BOTH halves will be enqueued on it right after the hook returns.  The hook enqueues the
 * transfer on that stream (ncclAllGather(..., stream)), or makes the stream wait for it (hipStreamWaitEvent), and returns at once:
 * non-zero = enqueued, 0 aborts the run.  It must also order the peer's read of this engine's half before this stream's NEXT UNet
 * call, which overwrites it (an all-gather on the stream does).  With this hook the denoise loop never drains the pipeline; with
 * sde_exchange_fn it does so once per step.  If both are set this one is used. */
typedef int (*sde_exchange_stream_fn)(float* d_model_out, size_t half_elems, int my_half, void* hip_stream, void* user);

typedef struct sde_txt2img_opts {
    int height, width;          /* must be divisible by 8 (impl_txt2img.rs:251-253) */
    float guidance_scale;       /* CFG is on iff > 1.0 (:258) */
    int has_rescale_cfg; float rescale_cfg; /* Option<f32> (:293-297) */
    int steps;
    int has_seed; uint64_t seed; /* None -> random (:248).  Latents come from StdRng::seed_from_u64(seed), scheduler noise from seed + 31337 (:249,267); see sde_rng_standard_normal. */
    int batch_size;             /* B = positive_prompt.len() (:256) */
    /* text embeddings from `encode_prompt` (impl_main.rs:339-375): f32 [N,S,D], N = 2B (uncond FIRST) when CFG is on, else B */
    const float* ehs; int S, D;
    const float* init_latents;  /* optional f32 [B,4,h,w] of N(0,1) draws replacing the seeded latents (:261-262) */
    const float* step_noise;    /* optional f32 [n_timesteps,B,4,h,w]: the scheduler's per-step N(0,1) draws */
    int buffers_on_device;      /* 1: ehs / init_latents / step_noise / images_out are DEVICE pointers */
    int skip_decode;            /* 1: return final latents only (images_out untouched) */
    int callback_kind, callback_frequency; sde_callback_fn callback; void* callback_user;
    /* optional taps for parity tests (host pointers, may be NULL) */
    float* noise_pred_out;      /* f32 [n_timesteps,B,4,h,w]: CFG-combined model output per step */
    float* latents_out;         /* f32 [n_timesteps,B,4,h,w]: latents after each scheduler step */
    float* final_latents_out;   /* f32 [B,4,h,w] */
    /* CFG-pair split: 0 = both halves on this engine (default); 1 = this engine runs the unconditional half, 2 = the text
     * half (UNet batch B instead of 2B).  `ehs` stays [2B,S,D]; seeds / init_latents / step_noise must be identical on both
     * engines (the scheduler step is replicated, so both end with the same latents).  Needs guidance_scale > 1.
     * Each engine decodes its share of the batch: rows [0, ceil(B/2)) of images_out on the unconditional engine, rows
     * [ceil(B/2), B) on the text engine; the other rows are left untouched. */
    int cfg_half; sde_exchange_fn exchange; void* exchange_user;
    /* Row capacities of the optional per-timestep buffers.  The number of timesteps is the scheduler's, not `steps`
     * (DDIM: steps - 1, ddim.rs:184; DDPM: ceil(T / (T / steps)), ddpm.rs:181 -- e.g. 167 for steps = 150), so a caller
     * sizes them from sde_scheduler_num_timesteps after sde_scheduler_set_timesteps(steps) and states the size here:
     * the run is rejected (SDE_ERR_INVALID_ARGUMENT) before anything is written when a buffer is too short.
     * 0 = not stated (then `steps` rows are assumed and a scheduler that yields more timesteps is rejected). */
    int trace_rows;             /* rows of noise_pred_out / latents_out */
    int step_noise_rows;        /* rows of step_noise */
    /* image-to-image start (no `run` for it exists in the reference, impl_img2img.rs; the arithmetic is diffusers 0.14.0's
     * StableDiffusionImg2ImgPipeline): the loop runs timesteps[start_step ..] only and `init_latents` (required when start_step > 0)
     * are taken as they are -- already noised by sde_scheduler_add_noise(0.18215 * encoded image, noise, timesteps[start_step]) --
     * instead of being multiplied by init_noise_sigma.  Callback / trace indices stay absolute (first call has step = start_step). */
    int start_step;
    sde_exchange_stream_fn exchange_stream;   /* CFG-pair split, stream-ordered exchange (above); shares exchange_user */
} sde_txt2img_opts;

void sde_txt2img_opts_default(sde_txt2img_opts* o);

typedef struct sde_run_stats {
    int unet_calls; float total_ms, unet_ms, vae_ms;   /* HIP-event times on the engine's stream */
    float host_enqueue_ms;   /* host wall time spent enqueueing the UNet calls (no waiting): << unet_ms means the loop is not launch-bound */
    int unet_graph_launches; /* UNet calls that went out as one hipGraph launch (the first call of a run, profiled passes and an engine whose
                              * capture failed go out launch by launch: a failed capture is not an error, the run completes on plain launches) */
} sde_run_stats;

/* Runs the whole loop on the device; `scheduler` is left in the post-run state the reference
 * leaves it in.  images_out: f32 [B,H,W,3] in [0,1]. */
sde_status sde_txt2img_run(sde_engine* e, sde_scheduler* scheduler, const sde_txt2img_opts* opts,
                           float* images_out, sde_run_stats* stats);

/* Seeded draws (host code; no device needed) --------------------------------------------------
 * What `Array::<f32>::random_using(shape, StandardNormal, &mut StdRng::seed_from_u64(seed))` yields in the reference
 * (impl_txt2img.rs:248-249,261-262,267; euler_ancestral_discrete.rs:203; ddpm.rs:224): ChaCha12 keyed by rand_core's
 * PCG32 seed expansion, 256-layer Ziggurat on f64, cast to f32, row-major.  `skip` samples are drawn and dropped first
 * (the scheduler stream after k steps of B*4*h*w draws).  The crates that define this (rand 0.8, rand_chacha 0.3,
 * rand_distr 0.4 via ndarray-rand 0.14, Cargo.toml:23) are not vendored by the reference and no known-answer stream of
 * theirs is available here: parity with the Rust stream is unpinned above the ChaCha core. */
sde_status sde_rng_standard_normal(uint64_t seed, uint64_t skip, float* out, size_t n);
/* The raw generator under it, for known-answer tests: `n_words` u32 outputs of ChaCha with `rounds` rounds (StdRng: 12)
 * under `key`, block counter and stream id starting at 0; and the key `seed_from_u64(seed)` derives. */
sde_status sde_rng_chacha_words(const uint8_t key[32], int rounds, uint32_t* out, size_t n_words);
sde_status sde_rng_seed_key(uint64_t seed, uint8_t key_out[32]);

/* Measurement hooks (bench.py) ------------------------------------------------------------- */
typedef struct sde_kernel_stat {
    char name[48];
    uint64_t launches;
    double total_ms;        /* HIP-event time summed over launches (on the engine's stream) */
    double flops;           /* algorithmic FLOPs (2*MAC) summed over launches */
    double bytes;           /* algorithmic bytes summed over launches */
} sde_kernel_stat;
/* enable=1: bracket every kernel launch with HIP events on the engine stream and accumulate per kernel class. */
sde_status sde_profile_enable(sde_engine* e, int enable);
sde_status sde_profile_reset(sde_engine* e);
/* Resolves pending events and copies up to `capacity` entries; *n_out = number of classes. */
sde_status sde_profile_get(sde_engine* e, sde_kernel_stat* out, int capacity, int* n_out);
sde_status sde_synchronize(sde_engine* e);

/* Kernel-level test hooks: run ONE contraction through the MFMA path (naive=0) or the plain path
 * (naive=1; 2 / 3 force the 256x320 / 256x160 kernels, 4 / 5 the 128x160 linear / halo 3x3 kernels, 6 the W-stationary streaming
 * kernel for short-K linears, gemm_ws.hip; sde_test_conv and sde_test_attention also take 7 / 8 = the fp32 engine's MFMA-tile /
 * plain kernels of f32path.hip, on unrounded fp32 operands) on host data.  conv: x NHWC f32 [NB,Hi,Wi,Cin], w OIHW f32, out NHWC f32.
 * upsample: 0 none, 1 nearest 2x before the conv, 2 nearest to (2*Hi-1) x (2*Wi-1) (the exported UNet's upsample to its skip
 * connection's size on odd latents), 3 = no upsample but the VAE encoder's downsampler window (pad (0,1,0,1), 3x3, stride 2, pad 0:
 * Ho = (Hi - 2) / 2 + 1).
 * out_capacity: the number of floats `out` holds.  The hook computes the output shape itself ([NB,Ho,Wo,Cout] resp. [NB,Sq,heads*d])
 * and refuses (SDE_ERR_INVALID_ARGUMENT, before anything is copied) unless that is exactly out_capacity -- a caller whose shape
 * arithmetic disagrees with the library's gets an error, never a write past its buffer. */
sde_status sde_test_conv(sde_engine* e, int naive, const float* x, int NB, int Hi, int Wi, int Cin,
                         const float* w, const float* bias, int Cout, int ksize, int stride, int upsample,
                         float* out, size_t out_capacity);
/* As sde_test_conv (stride 1, no upsample; naive also takes 4 / 5 = the 128x160 linear kernel / the halo 3x3 kernel) with a
 * residual added in the epilogue: residual NHWC f32 [NB,Hi,Wi,Cout].  split = 0: residual and result are plain fp16 tensors.
 * split = 1: the residual goes in as a split-fp16 pair (hi + lo) and the result comes back as one (out = hi + lo), the form the
 * UNet trunk uses (GemmArgs::Rlo / Dlo).  Test infrastructure; no counterpart in the reference. */
sde_status sde_test_conv_res(sde_engine* e, int naive, const float* x, int NB, int Hi, int Wi, int Cin,
                             const float* w, const float* bias, int Cout, int ksize, const float* residual, int split,
                             float* out, size_t out_capacity);
/* The GEGLU projection (BiasSplitGelu, scripts/_optimizer/fusion_biassplitgelu.py:15-110) as one GEMM with the gate product in the
 * epilogue: x f32 [M,K]; w f32 [N,K] with the N/2 value rows first, then the N/2 gate rows (file order); out f32 [M,N/2] =
 * (x Wv^T + bv) * gelu(x Wg^T + bg), exact-erf GELU or (gelu_tanh = 1) FastGelu's tanh form.  naive: 0 heuristic, 1 plain, 2 the
 * 256x320 kernel, 6 the W-stationary streaming kernel. */
sde_status sde_test_geglu(sde_engine* e, int naive, const float* x, int M, int K, const float* w, const float* bias, int N,
                          int gelu_tanh, float* out, size_t out_capacity);
sde_status sde_test_attention(sde_engine* e, int naive, const float* q, const float* k, const float* v,
                              int NB, int heads, int Sq, int Skv, int d, float* out, size_t out_capacity);

#ifdef __cplusplus
}
#endif
#endif /* SDENGINE_H */
