// kernels.h — launch interface of the hand-written gfx950 kernels.
//
// Layout conventions inside the engine:
//   activations  : fp16, NHWC == [rows = NB*H*W][C] row-major ("token-major"), C % 8 == 0
//   weights      : fp16, [Npad][K] row-major, K = taps*Cin.  3x3: K index = ((cin/64)*9 + tap)*64 + cin%64 when Cin % 64 == 0
//                  (chunk-major: the 9 taps of a 64-channel chunk are consecutive K-tiles), else tap*Cin + cin
//   accumulators : fp32
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------ profiling
struct Profiler;  // engine.cpp
struct LaunchCtx {
    hipStream_t stream = nullptr;
    Profiler* prof = nullptr;
    const half_t* zeros = nullptr;  // >= 256 B of device zeros (source for padded taps / tail rows)
    bool force_naive = false;
    bool force_mid = false;         // test hook: take the 8-wave 256x160 kernel whenever it supports the problem
    bool force_mid128 = false;      // test hook: take the 128x160 loader-wave kernel (linears) whenever it supports the problem
    bool force_midh = false;        // test hook: take the halo 3x3 kernel (16 x 16 output patches) whenever it supports the problem
    bool force_big = false;         // test hook: take the 8-wave 256x320 kernel whenever it supports the problem
    bool force_ws = false;          // test hook: take the W-stationary streaming kernel (gemm_ws.hip) whenever it supports the problem
    float* splitk_ws = nullptr;     // fp32 partial-tile workspace for split-K GEMMs (null: split-K off)
    size_t splitk_ws_bytes = 0;
    int splitk_force = 0;           // tuning hook: >0 forces this split factor where split-K applies
    // fp32 engine (sde_engine_opts.dtype = SDE_F32): every activation / weight pointer declared `half_t*` below addresses FLOATS and
    // every launcher forwards to its fp32 kernel in f32path.hip.  Element counts, strides and layouts are unchanged.
    bool f32 = false;
    size_t esize() const { return f32 ? 4 : 2; }
};
// f32path.hip
struct GemmArgs; struct AttnArgs; struct GroupNormArgs;
void f32_launch_gemm(const LaunchCtx& ctx, const GemmArgs& a);
void f32_launch_geglu(const LaunchCtx& ctx, const half_t* X, half_t* Y, int64_t rows, int half_n, int tanh_form);
void f32_launch_attention(const LaunchCtx& ctx, const AttnArgs& a);
void f32_launch_groupnorm(const LaunchCtx& ctx, const GroupNormArgs& a);
void f32_launch_layernorm_sum(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, float* S32_out, half_t* Y, int rows, int C,
                              const float* gamma, const float* beta, float eps);
void f32_launch_sum_round(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, half_t* Y, int64_t n);
void f32_launch_pack_nchw(const LaunchCtx& ctx, const float* x, half_t* y, int NB, int C, int H, int W, int Cpad, float scale);
void f32_launch_pack_latents(const LaunchCtx& ctx, const float* latents, half_t* packed, int B, int dup, int C, int H, int W, int Cpad, float scale);
void f32_launch_gemv(const LaunchCtx& ctx, const half_t* W, const float* x, const float* bias, float* y, int N, int K, bool pre_silu, bool post_silu);
// RAII bracket: records HIP events around a launch when profiling is on.
// Raises a kernel's dynamic-LDS limit once per (device, kernel): the attribute is per device, and engines on several GPUs (or on
// several host threads) may share this process.
void allow_dynamic_lds(const void* kernel, int bytes);

struct ProfScope {
    ProfScope(const LaunchCtx& ctx, const char* name, double flops, double bytes);
    ~ProfScope();
    const LaunchCtx& ctx;
    int slot;
};

// ------------------------------------------------------------------------------------------ GEMM / implicit-GEMM conv
enum GemmEpilogue {
    EPI_STD = 0,       // D[m][n] = half(alpha*acc + bias[n] + R[m][n])
    EPI_GEGLU = 1,     // weights row-interleaved in 16-blocks (value,gate); D[m][j] = (a+ba) * gelu_erf(g+bg), ldd >= N/2
    EPI_NCHW_F32 = 2,  // Df[(nb*N + n)*HoWo + pix] = acc + bias          (UNet / VAE graph output)
    EPI_IMAGE = 3,     // Df[m*N + n] = clamp((acc + bias)/2 + 0.5, 0, 1)   (decode_latents post-processing)
    EPI_TRANS = 4,     // D[n*ldd + m] = half(alpha*acc + bias[n])
    EPI_GEGLU_PAIR = 5 // as EPI_GEGLU with weight rows interleaved in fours: rows 4q..4q+3 = value_2q, value_2q+1, gate_2q, gate_2q+1 (gemm_big.hip)
};

struct GemmArgs {
    // A operand: rows m in [0,M); for conv, m = (nb*Ho + oy)*Wo + ox and K runs over (tap, channel)
    const half_t* A0 = nullptr;
    const half_t* A1 = nullptr;  // second source of a channel concat (or null)
    int C0 = 0, C1 = 0;          // channels taken from A0 / A1 per pixel (C0 + C1 = Cin)
    int lda0 = 0, lda1 = 0;      // row strides in elements
    int M = 0, N = 0, K = 0;     // K = taps * (C0 + C1)
    int taps = 1;                // 1 (linear / 1x1) or 9 (3x3, pad 1)
    int stride = 1;              // conv stride (1 or 2)
    int coff = 0;                // window centre = o*stride + coff: 0 = symmetric pad 1; 1 = the VAE encoder's pad (0,1,0,1) + stride-2, pad-0 conv
    int ups = 0;                 // 1: the source tensor is (Hi/2 x Wi/2), nearest-upsampled 2x on the fly
    int Hi = 0, Wi = 0;          // (virtual) input height/width; Ho, Wo output
    int Ho = 0, Wo = 0;
    const half_t* W = nullptr;   // [Npad][K]
    int Npad = 0;
    const float* bias = nullptr; // [N] or null
    const half_t* R = nullptr;   // residual [M][ldr] or null
    // Split-fp16 tensors (EPI_STD only): a tensor that is never an MFMA operand is kept as hi + lo, hi = half(x), lo = half(x - hi)
    // (~22 significant bits, same layout and stride as its hi part).  Rlo: low part of the residual; Dlo: where to store the
    // low part of the result.  Consumers that need an MFMA operand read hi alone.  DESIGN.md section 2.
    const half_t* Rlo = nullptr;
    half_t* Dlo = nullptr;
    int ldr = 0;
    half_t* D = nullptr;
    float* Df = nullptr;
    int ldd = 0;
    int epi = EPI_STD;
    int gelu_tanh = 0;           // GEGLU epilogues: 1 = com.microsoft.FastGelu's tanh form (optimizer-fused models), 0 = exact erf
    float alpha = 1.0f;
    // batching over blockIdx.z (independent problems with identical shapes)
    int batch = 1;
    long long sA = 0, sW = 0, sD = 0, sR = 0;
    const char* tag = "gemm";
};

void launch_gemm(const LaunchCtx& ctx, const GemmArgs& a);
// true if the MFMA path can take this problem (alignment rules in gemm.hip)
bool gemm_mfma_supported(const GemmArgs& a);
int gemm_npad(int N);  // weight rows must be padded to this

// ------------------------------------------------------------------------------------------ attention
struct AttnArgs {
    const half_t* Q = nullptr; const half_t* K = nullptr; const half_t* V = nullptr; half_t* O = nullptr;
    int ldq = 0, ldk = 0, ldv = 0, ldo = 0;  // row strides (elements)
    int NB = 0, heads = 0, Sq = 0, Skv = 0, d = 0;
    float scale = 1.0f;
};
void launch_attention(const LaunchCtx& ctx, const AttnArgs& a);
bool attention_mfma_supported(const AttnArgs& a);

// ------------------------------------------------------------------------------------------ normalisation
// GroupNorm over NHWC (optionally a channel concat of two sources), fused affine (+SiLU).
struct GroupNormArgs {
    const half_t* X0 = nullptr; const half_t* X1 = nullptr;
    const half_t* X0lo = nullptr; const half_t* X1lo = nullptr;   // low parts of split-fp16 inputs (or null): added before normalising
    int C0 = 0, C1 = 0;
    int NB = 0, HW = 0, groups = 32;
    float eps = 1e-5f;
    const float* gamma = nullptr; const float* beta = nullptr;  // [C0+C1] fp32
    bool silu = false;
    half_t* Y = nullptr;     // [NB*HW][C0+C1]
    float* ws = nullptr;     // workspace, >= groupnorm_ws_floats(...) floats
};
size_t groupnorm_ws_floats(int NB, int C, int HW);
void launch_groupnorm(const LaunchCtx& ctx, const GroupNormArgs& a);

void launch_layernorm(const LaunchCtx& ctx, const half_t* X, half_t* Y, int rows, int C, const float* gamma,
                      const float* beta, float eps);
// y = LayerNorm(S32 + A + B) with the sum formed and kept in fp32 (any of the three may be null, not all); S32_out (optional) receives the sum
void launch_layernorm_sum(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, float* S32_out, half_t* Y, int rows, int C,
                          const float* gamma, const float* beta, float eps);
// fp16 engine: y = LayerNorm((A + B) + C) with the sum in fp32 registers only (C may be null); Y4 = half(((A + B) + C) + D) likewise
void launch_layernorm_sum3(const LaunchCtx& ctx, const half_t* A, const half_t* B, const half_t* Cc, half_t* Y, int rows, int C, const float* gamma,
                           const float* beta, float eps);
void launch_sum_round4(const LaunchCtx& ctx, const half_t* A, const half_t* B, const half_t* Cc, const half_t* Dd, half_t* Y, half_t* Ylo, int64_t n);
// Y = half(S32 + A + B), n elements (n % 8 == 0); Ylo (optional) receives the low part half(sum - Y) of a split-fp16 result
void launch_sum_round(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, half_t* Y, half_t* Ylo, int64_t n);
// in-place softmax over the last dim of a [rows][cols] fp16 matrix (fp32 math)
void launch_softmax_rows(const LaunchCtx& ctx, half_t* X, int64_t rows, int cols, int pitch);   // in place; columns [cols, pitch) become 0

// ------------------------------------------------------------------------------------------ element-wise / edge kernels
// NCHW f32 [NB,C,H,W] -> NHWC fp16 [NB*H*W][Cpad] (channels >= C zero), scaled
void launch_pack_nchw(const LaunchCtx& ctx, const float* x, half_t* y, int NB, int C, int H, int W, int Cpad, float scale);
void launch_f32_to_f16(const LaunchCtx& ctx, const float* x, half_t* y, int64_t n);
void launch_f16_to_f32(const LaunchCtx& ctx, const half_t* x, float* y, int64_t n);
// NHWC fp16 -> NHWC f32 (test hook)
// sinusoidal timestep embedding, flip_sin_to_cos=True, shift 0: out[0:dim/2]=cos, [dim/2:]=sin  (f32)
// t is read from DEVICE memory (one float): a UNet call captured into a hipGraph replays with whatever timestep the scalar holds
void launch_timestep_embedding(const LaunchCtx& ctx, const float* t_dev, float* out, int dim);
void launch_set_scalar(const LaunchCtx& ctx, float* dst, float value);
// y[n] = act(W[n][:] . x + bias[n]),  W fp16 [N][K], x f32 [K]; pre_silu applies SiLU to x first
void launch_gemv(const LaunchCtx& ctx, const half_t* W, const float* x, const float* bias, float* y, int N, int K,
                 bool pre_silu, bool post_silu);
// approximate_decode_latents: latents NCHW f32 [B,4,h,w] -> NHWC f32 [B,h,w,3] clamped
void launch_approx_decode(const LaunchCtx& ctx, const float* lat, float* img, int B, int h, int w);

// VAE posterior: moments f32 NCHW [B, 2*LC, h, w] (mean | logvar) -> out f32 [B, LC, h, w] = mean + exp(0.5 clamp(logvar)) * noise (noise null: mean)
void launch_vae_sample(const LaunchCtx& ctx, const float* moments, const float* noise, float* out, int B, int latent_channels, int h, int w);

// weight repack: src (f32 or f16) -> dst fp16 [rows][Kdst], see loader for the index maps
struct RepackArgs {
    const void* src = nullptr; int src_is_f16 = 0;
    // logical source tensor W[o][i][ky][kx] (conv, OIHW) or W[o][i] (linear);  in_out=1: stored [i][o]
    int O = 0, I = 0, KH = 1, KW = 1, in_out = 0;
    half_t* dst = nullptr; int dst_row0 = 0;  // destination row offset
    int Ipad = 0;                              // per-tap channel count in dst (>= I, zero padded)
    int i0 = 0;                                // channel offset inside the tap where this source starts
    int Kdst = 0;                              // dst row length
    int geglu_half = 0;                        // >0: row o -> interleaved row (16-block value/gate), value = geglu_half
    int geglu_pair = 0;                        // with geglu_half: interleave in fours instead (value_j -> 4(j/2) + j%2, gate_j -> that + 2)
    int src_i0 = 0, src_I = 0;                 // copy only source channels [src_i0, src_i0 + src_I) (src_I = 0: all I of them)
    int dst_f32 = 0;                           // fp32 engine: dst is float [rows][Kdst]
};
void launch_repack(const LaunchCtx& ctx, const RepackArgs& a);

// ------------------------------------------------------------------------------------------ fused CFG + scheduler step
enum StepKind { STEP_EULER = 0, STEP_EULER_A = 1, STEP_DPM1 = 2, STEP_DPM2M = 3, STEP_DDIM = 4, STEP_DDPM = 5, STEP_LMS = 6, STEP_DPM3M = 7 };
struct StepArgs {
    int kind = 0;
    int pred = 0;            // sde_prediction_type
    int B = 0, C = 4, H = 0, W = 0;
    int cfg = 0;             // 1: model_out holds [2B] (uncond first)
    float guidance = 1.0f;
    const float* rescale_factor = nullptr;  // device scalar (m*ro_pos/ro_cfg + 1-m) or null
    const float* model_out = nullptr;       // f32 NCHW [N,4,h,w]
    float* latents = nullptr;               // f32 NCHW [B,4,h,w], updated in place
    const float* noise = nullptr;           // f32 [B,4,h,w] or null
    float c[12] = {0};                      // scheduler coefficients (meaning per kind, scheduler.cpp)
    int clip = 0;
    int dpm_algo = 0;                       // sde_dpm_algorithm: how DPM-Solver converts the model output (dpm_solver_multistep.rs:211-246)
    float* hist[4] = {nullptr, nullptr, nullptr, nullptr};  // history buffers (DPM multistep: converted outputs; LMS: derivatives), f32 [B,4,h,w]
    int nhist = 0;
    float lms_c[4] = {0};
    float next_input_scale = 1.0f;          // next step's scale_model_input factor (1/sqrt(sigma^2+1) or 1)
    half_t* next_input = nullptr;           // fp16 (fp32 engine: float) NHWC [N][h*w][Cpad] for the next UNet call (or null)
    int next_input_f32 = 0;
    int Cpad = 64;
    float* eps_out = nullptr;               // optional f32 [B,4,h,w] tap: CFG-combined model output
    float* x0_out = nullptr;                // optional f32 [B,4,h,w]: SchedulerStepOutput::pred_original_sample (schedulers/mod.rs:111) where the step has one
    float* deriv_out = nullptr;             // optional f32 [B,4,h,w]: the Euler family's derivative (euler_discrete.rs:210)
};
void launch_cfg_step(const LaunchCtx& ctx, const StepArgs& a);
// latents -> scaled, duplicated, packed UNet input (first step / standalone scale_model_input)
void launch_pack_latents(const LaunchCtx& ctx, const float* latents, half_t* packed, int B, int dup, int C, int H, int W,
                         int Cpad, float scale);
// CFG-rescale factor (impl_txt2img.rs:293-297): out[0] = m*std(text)/std(x_cfg) + (1-m)
void launch_cfg_rescale_factor(const LaunchCtx& ctx, const float* model_out, int B, int64_t per_half, float guidance,
                               float m, float* ws, float* out);
// mode 0: out = x / a ; mode 1: out = a*x + b*y ; mode 2: out = x * a   (f32 vectors)
void launch_vec_op(const LaunchCtx& ctx, const float* x, const float* y, float* out, int64_t n, int mode, float a, float b);
