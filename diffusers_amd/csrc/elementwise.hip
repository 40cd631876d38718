// elementwise.hip — edge / element-wise kernels: layout packing, timestep embedding, GEMV for the
// time-embedding MLP, weight repack, and the fused classifier-free-guidance + scheduler-step kernel
// (DESIGN.md §4.4).  All HBM- or latency-bound.
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------ packing / casts
__global__ void pack_nchw_kernel(const float* x, half_t* y, int NB, int C, int HW, int Cpad, float scale)
{
    long long total = (long long)NB * HW * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int c = (int)(i % Cpad);
        long long row = i / Cpad;
        int nb = (int)(row / HW), pix = (int)(row - (long long)nb * HW);
        float v = c < C ? x[((long long)nb * C + c) * HW + pix] * scale : 0.f;
        y[i] = (half_t)v;
    }
}
void launch_pack_nchw(const LaunchCtx& ctx, const float* x, half_t* y, int NB, int C, int H, int W, int Cpad, float scale)
{
    if (ctx.f32) { f32_launch_pack_nchw(ctx, x, y, NB, C, H, W, Cpad, scale); return; }
    ProfScope ps(ctx, "pack_nchw", 0, (double)NB * H * W * (C * 4.0 + Cpad * 2.0));
    long long total = (long long)NB * H * W * Cpad;
    int blocks = (int)std::min<long long>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(pack_nchw_kernel, dim3(blocks), dim3(256), 0, ctx.stream, x, y, NB, C, H * W, Cpad, scale);
    HIP_CHECK(hipGetLastError());
}

__global__ void pack_latents_kernel(const float* lat, half_t* packed, int B, int dup, int C, int HW, int Cpad, float scale)
{
    // writes only channels < C; the pad channels of `packed` are zeroed once at allocation
    long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int pix = (int)(i % HW);
        int c = (int)((i / HW) % C);
        int b = (int)(i / ((long long)HW * C));
        half_t v = (half_t)(lat[i] * scale);
        for (int r = 0; r < dup; ++r) packed[(((long long)(b + r * B)) * HW + pix) * Cpad + c] = v;
    }
}
void launch_pack_latents(const LaunchCtx& ctx, const float* latents, half_t* packed, int B, int dup, int C, int H, int W, int Cpad,
                         float scale)
{
    if (ctx.f32) { f32_launch_pack_latents(ctx, latents, packed, B, dup, C, H, W, Cpad, scale); return; }
    ProfScope ps(ctx, "pack_latents", 0, (double)B * C * H * W * (4.0 + 2.0 * dup));
    long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(pack_latents_kernel, dim3((unsigned)std::min<long long>(cdiv64(total, 256), 2048)), dim3(256), 0, ctx.stream,
                       latents, packed, B, dup, C, H * W, Cpad, scale);
    HIP_CHECK(hipGetLastError());
}

__global__ void f32_to_f16_kernel(const float* x, half_t* y, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = (half_t)x[i];
}
__global__ void f16_to_f32_kernel(const half_t* x, float* y, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = (float)x[i];
}
void launch_f32_to_f16(const LaunchCtx& ctx, const float* x, half_t* y, int64_t n)
{
    if (ctx.f32) { HIP_CHECK(hipMemcpyAsync(y, x, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx.stream)); return; }   // fp32 engine: a copy
    ProfScope ps(ctx, "cast_f32_f16", 0, 6.0 * n);
    hipLaunchKernelGGL(f32_to_f16_kernel, dim3((unsigned)std::min<long long>(cdiv64(n, 256), 4096)), dim3(256), 0, ctx.stream, x, y, (long long)n);
    HIP_CHECK(hipGetLastError());
}
void launch_f16_to_f32(const LaunchCtx& ctx, const half_t* x, float* y, int64_t n)
{
    ProfScope ps(ctx, "cast_f16_f32", 0, 6.0 * n);
    hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)std::min<long long>(cdiv64(n, 256), 4096)), dim3(256), 0, ctx.stream, x, y, (long long)n);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ time embedding
// Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): emb = t * exp(-ln(10000) * k / (dim/2)); [cos | sin]
__global__ void set_scalar_kernel(float* dst, float v) { if (threadIdx.x == 0) *dst = v; }
void launch_set_scalar(const LaunchCtx& ctx, float* dst, float value)
{
    hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(64), 0, ctx.stream, dst, value);
    HIP_CHECK(hipGetLastError());
}

__global__ void timestep_embedding_kernel(const float* t_dev, float* out, int dim)
{
    const float t = *t_dev;
    int half_dim = dim / 2;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= half_dim) return;
    float freq = expf(-9.210340371976184f * (float)k / (float)half_dim);
    float a = t * freq;
    out[k] = cosf(a);
    out[half_dim + k] = sinf(a);
}
void launch_timestep_embedding(const LaunchCtx& ctx, const float* t_dev, float* out, int dim)
{
    ProfScope ps(ctx, "timestep_embedding", 0, 4.0 * dim);
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(cdiv(dim / 2, 64)), dim3(64), 0, ctx.stream, t_dev, out, dim);
    HIP_CHECK(hipGetLastError());
}

// one wave per output row
__global__ void gemv_kernel(const half_t* W, const float* x, const float* bias, float* y, int N, int K, int pre_silu, int post_silu)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    const half_t* w = W + (long long)n * K;
    float acc = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        half8 wv = *(const half8*)(w + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float xv = x[k + j];
            if (pre_silu) xv = xv / (1.0f + __expf(-xv));
            acc += (float)wv[j] * xv;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        float v = acc + (bias ? bias[n] : 0.f);
        if (post_silu) v = v / (1.0f + __expf(-v));
        y[n] = v;
    }
}
void launch_gemv(const LaunchCtx& ctx, const half_t* W, const float* x, const float* bias, float* y, int N, int K, bool pre_silu,
                 bool post_silu)
{
    if (ctx.f32) { f32_launch_gemv(ctx, W, x, bias, y, N, K, pre_silu, post_silu); return; }
    SDE_REQUIRE(K % 8 == 0, SDE_ERR_INVALID_ARGUMENT, "gemv: K must be a multiple of 8");
    ProfScope ps(ctx, "gemv_time_mlp", 2.0 * N * K, 2.0 * N * K);
    hipLaunchKernelGGL(gemv_kernel, dim3(cdiv(N, 4)), dim3(256), 0, ctx.stream, W, x, bias, y, N, K, pre_silu ? 1 : 0, post_silu ? 1 : 0);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ VAE posterior sample
// DiagonalGaussianDistribution.sample() as the exported vae_encoder.onnx computes it (sd2pyke.py:181-187): moments = (mean | logvar)
// along channels, logvar clamped to [-30, 20], sample = mean + exp(0.5 logvar) * noise; noise == null returns the mean (mode()).
__global__ void vae_sample_kernel(const float* moments, const float* noise, float* out, int B, int LCHW)
{
    const long long total = (long long)B * LCHW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / LCHW, r = i - b * LCHW;
        const float mean = moments[b * 2 * LCHW + r];
        const float logvar = fminf(fmaxf(moments[b * 2 * LCHW + LCHW + r], -30.0f), 20.0f);
        out[i] = noise ? mean + expf(0.5f * logvar) * noise[i] : mean;
    }
}
void launch_vae_sample(const LaunchCtx& ctx, const float* moments, const float* noise, float* out, int B, int latent_channels, int h, int w)
{
    const long long total = (long long)B * latent_channels * h * w;
    ProfScope ps(ctx, "vae_posterior_sample", 0, 16.0 * total);
    hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)std::min<long long>(cdiv64(total, 256), 4096)), dim3(256), 0, ctx.stream, moments, noise, out, B,
                       latent_channels * h * w);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ approximate decode
// impl_main.rs:385-395: einsum("blxy,lr->bxyr") with a fixed 4x3 matrix, then clamp [0,1] (to_image).
__global__ void approx_decode_kernel(const float* lat, float* img, int B, int HW)
{
    const float coef[4][3] = {{0.298f, 0.207f, 0.208f}, {0.187f, 0.286f, 0.173f}, {-0.158f, 0.189f, 0.264f}, {-0.184f, -0.271f, -0.473f}};
    long long total = (long long)B * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int b = (int)(i / HW), pix = (int)(i - (long long)b * HW);
        float l[4];
        for (int c = 0; c < 4; ++c) l[c] = lat[((long long)b * 4 + c) * HW + pix];
        for (int r = 0; r < 3; ++r) {
            float v = 0.f;
            for (int c = 0; c < 4; ++c) v += l[c] * coef[c][r];
            img[i * 3 + r] = fminf(fmaxf(v, 0.f), 1.f);
        }
    }
}
void launch_approx_decode(const LaunchCtx& ctx, const float* lat, float* img, int B, int h, int w)
{
    ProfScope ps(ctx, "approx_decode", 0, (double)B * h * w * 28.0);
    hipLaunchKernelGGL(approx_decode_kernel, dim3((unsigned)cdiv64((long long)B * h * w, 256)), dim3(256), 0, ctx.stream, lat, img, B, h * w);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ weight repack
struct RepackDev {
    const void* src; int src_is_f16;
    int O, I, KH, KW, in_out;
    half_t* dst; int dst_row0, Ipad, i0, Kdst, geglu_half, geglu_pair, src_i0, nI, dst_f32;
};
__global__ void repack_kernel(RepackDev a)
{
    long long total = (long long)a.O * a.nI * a.KH * a.KW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        // idx enumerates the DESTINATION order (o, tap, i) so that writes coalesce; i counts from the first copied source channel
        int i = (int)(idx % a.nI);
        long long t = idx / a.nI;
        int tap = (int)(t % (a.KH * a.KW));
        int o = (int)(t / (a.KH * a.KW));
        const int is = a.src_i0 + i;                                                // source channel
        long long s;
        if (a.in_out) s = (long long)is * a.O + o;                                  // [I][O] (onnx::MatMul_n)
        else s = (((long long)o * a.I + is) * a.KH * a.KW) + tap;                   // OIHW / [O][I]
        float v = a.src_is_f16 ? (float)((const half_t*)a.src)[s] : ((const float*)a.src)[s];
        int row = o;
        if (a.geglu_half > 0) {
            int j = o < a.geglu_half ? o : o - a.geglu_half;
            row = a.geglu_pair ? 4 * (j >> 1) + (j & 1) + (o < a.geglu_half ? 0 : 2) : 32 * (j >> 4) + (j & 15) + (o < a.geglu_half ? 0 : 16);
        }
        // K order of a 3x3 weight row: 64-channel chunks outermost, then the 9 taps, then the channel inside the chunk
        // (all taps of a chunk are consecutive K-tiles, so the im2col re-reads of an activation line hit L2; kernels.h)
        const int c = a.i0 + i, taps = a.KH * a.KW;
        const int ck = (taps > 1 && a.Ipad % 64 == 0) ? 64 : a.Ipad;
        const long long kcol = ((long long)(c / ck) * taps + tap) * ck + c % ck;
        if (a.dst_f32) ((float*)a.dst)[(long long)(a.dst_row0 + row) * a.Kdst + kcol] = v;
        else a.dst[(long long)(a.dst_row0 + row) * a.Kdst + kcol] = (half_t)v;
    }
}
void launch_repack(const LaunchCtx& ctx, const RepackArgs& r)
{
    const int nI = r.src_I > 0 ? r.src_I : r.I;
    SDE_REQUIRE(r.src_i0 >= 0 && r.src_i0 + nI <= r.I, SDE_ERR_INVALID_ARGUMENT, "repack: source channel range");
    RepackDev a{r.src, r.src_is_f16, r.O, r.I, r.KH, r.KW, r.in_out, r.dst, r.dst_row0, r.Ipad, r.i0, r.Kdst, r.geglu_half, r.geglu_pair, r.src_i0, nI, r.dst_f32};
    long long total = (long long)a.O * nI * a.KH * a.KW;
    hipLaunchKernelGGL(repack_kernel, dim3((unsigned)std::min<long long>(cdiv64(total, 256), 8192)), dim3(256), 0, ctx.stream, a);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ fused CFG + scheduler step
// One launch per denoise step does: classifier-free-guidance combine (impl_txt2img.rs:288-301), the scheduler's
// `step` (src/schedulers/*.rs) in the reference's f32 operation order, and the NEXT step's `scale_model_input`
// + concat + fp16 NHWC packing (impl_txt2img.rs:273-278), so latents never leave HBM.
// Algorithmic bytes per latent element: read uncond, text, sample (+noise) = 12-16 B, write sample 4 B + 2x2 B
// packed input = 20-24 B (+4 per history tensor).
struct StepDev {
    int kind, pred, B, C, HW, cfg;
    float guidance;
    const float* rescale_factor;
    const float* model_out; float* latents; const float* noise;
    float c[12]; int clip, dpm_algo;
    float* hist[4]; int nhist; float lms_c[4];
    float next_scale; half_t* next_input; int Cpad, next_f32;
    float* eps_out; float* x0_out; float* deriv_out;
};

__global__ void cfg_step_kernel(StepDev a)
{
    const long long per = (long long)a.B * a.C * a.HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        float e;
        if (a.cfg) {
            float u = a.model_out[i], t = a.model_out[per + i];
            e = u + a.guidance * (t - u);                                     // impl_txt2img.rs:299
            if (a.rescale_factor) {                                           // :293-297
                float f = a.rescale_factor[0], m = a.rescale_factor[1];
                e = m * (e * f) + (1.0f - m) * e;
            }
        } else e = a.model_out[i];
        if (a.eps_out) a.eps_out[i] = e;
        const float s = a.latents[i];
        float prev;
        switch (a.kind) {
        case STEP_EULER: {   // euler_discrete.rs:208-214 ; c0 = sigma_hat, c1 = dt
            float pred = s - a.c[0] * e;
            float d = (s - pred) / a.c[0];
            prev = s + d * a.c[1];
            if (a.x0_out) a.x0_out[i] = pred;
            if (a.deriv_out) a.deriv_out[i] = d;
        } break;
        case STEP_EULER_A: { // euler_ancestral_discrete.rs:194-204 ; c0 = sigma_from, c1 = dt, c2 = sigma_up
            float pred = s - a.c[0] * e;
            float d = (s - pred) / a.c[0];
            prev = s + d * a.c[1];
            prev = prev + a.noise[i] * a.c[2];
            if (a.x0_out) a.x0_out[i] = pred;
            if (a.deriv_out) a.deriv_out[i] = d;
        } break;
        case STEP_DPM1:
        case STEP_DPM2M:
        case STEP_DPM3M: {   // dpm_solver_multistep.rs:211-326 ; c0 = alpha_s, c1 = sigma_s (current timestep), c2 multiplies the sample,
                             // c3 multiplies the converted output m0 (D++: sigma_t/sigma_s, alpha_t*(exp(-h)-1); DPMSolver: alpha_t/alpha_s,
                             // sigma_t*(exp(h)-1), :254-255)
            float m0;
            if (a.dpm_algo == 0) {                     // DPMSolverPlusPlus: x0 prediction (:213-231)
                if (a.pred == SDE_PRED_EPSILON) m0 = (s - a.c[1] * e) / a.c[0];
                else if (a.pred == SDE_PRED_SAMPLE) m0 = e;
                else m0 = a.c[0] * s - a.c[1] * e;
            } else {                                   // DPMSolver: noise prediction (:232-244)
                if (a.pred == SDE_PRED_EPSILON) m0 = e;
                else if (a.pred == SDE_PRED_SAMPLE) m0 = (s - a.c[0] * e) / a.c[1];
                else m0 = a.c[0] * e + a.c[1] * s;
            }
            if (a.kind == STEP_DPM1) prev = a.c[2] * s - a.c[3] * m0;
            else if (a.kind == STEP_DPM2M) {
                // :259-285 ; c4 = 1/r0 ; hist[0] = m1.  c5 = the d1 coefficient as the reference SUBTRACTS it (Midpoint 0.5*c3;
                // DPMSolver Heun sigma_t*((exp(h)-1)/h - 1); D++ Heun adds its term, so the host passes it negated: x - (-c)*d == x + c*d)
                float m1 = a.hist[0][i];
                float d1 = a.c[4] * (m0 - m1);
                prev = a.c[2] * s - a.c[3] * m0 - a.c[5] * d1;
            } else {
                // :288-326 ; hist[0] = m1, hist[1] = m2 ; c4 = 1/r0, c6 = 1/r1, c7 = r0/(r0+r1), c8 = 1/(r0+r1), c5 / c9 = d1 / d2 coefficients
                // (c5 negated by the host for D++, whose d1 term is added)
                float m1 = a.hist[0][i], m2 = a.hist[1][i];
                float d10 = a.c[4] * (m0 - m1), d11 = a.c[6] * (m1 - m2);
                float d1 = d10 + a.c[7] * (d10 - d11);
                float d2 = a.c[8] * (d10 - d11);
                prev = a.c[2] * s - a.c[3] * m0 - a.c[5] * d1 - a.c[9] * d2;
            }
            if (a.nhist > 0) a.hist[a.nhist - 1][i] = m0;  // the host rotates the buffers after every step (write slot -> hist[0])
        } break;
        case STEP_DDIM: {    // ddim.rs:210-238 ; c0 = sqrt(alpha_t), c1 = sqrt(beta_t), c2 = sqrt(alpha_prev), c3 = sqrt(1 - alpha_prev - std^2)
            float mo = e, x0;
            if (a.pred == SDE_PRED_EPSILON) x0 = (s - a.c[1] * mo) / a.c[0];
            else if (a.pred == SDE_PRED_SAMPLE) x0 = mo;
            else { mo = a.c[0] * mo + a.c[1] * s; x0 = a.c[0] * s - a.c[1] * mo; }
            if (a.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            prev = a.c[2] * x0 + a.c[3] * mo;
            if (a.x0_out) a.x0_out[i] = x0;
        } break;
        case STEP_DDPM: {    // ddpm.rs:203-232 ; c0 = sqrt(alpha_t), c1 = sqrt(beta_t), c2 = x0 coeff, c3 = sample coeff, c4 = noise coeff (0 at pos 0)
            float x0;
            if (a.pred == SDE_PRED_EPSILON) x0 = (s - a.c[1] * e) / a.c[0];
            else x0 = e;
            if (a.clip) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
            prev = a.c[2] * x0 + a.c[3] * s;
            if (a.noise) prev = prev + a.c[4] * a.noise[i];
            if (a.x0_out) a.x0_out[i] = x0;
        } break;
        default: {           // STEP_LMS ; c0 = sigma ; lms_c[k] multiplies derivative (i-k); hist[0] receives the new derivative
            float pred = s - a.c[0] * e;
            float d = (s - pred) / a.c[0];
            prev = s + a.lms_c[0] * d;
            for (int k = 1; k < a.nhist; ++k) prev = prev + a.lms_c[k] * a.hist[k][i];
            a.hist[0][i] = d;
            if (a.x0_out) a.x0_out[i] = pred;
            if (a.deriv_out) a.deriv_out[i] = d;
        } break;
        }
        a.latents[i] = prev;
        if (a.next_input) {
            int pix = (int)(i % a.HW);
            int c = (int)((i / a.HW) % a.C);
            int b = (int)(i / ((long long)a.HW * a.C));
            if (a.next_f32) {
                float* ni = (float*)a.next_input;
                const float v = prev * a.next_scale;
                ni[((long long)b * a.HW + pix) * a.Cpad + c] = v;
                if (a.cfg) ni[((long long)(b + a.B) * a.HW + pix) * a.Cpad + c] = v;
            } else {
                half_t v = (half_t)(prev * a.next_scale);
                a.next_input[((long long)b * a.HW + pix) * a.Cpad + c] = v;
                if (a.cfg) a.next_input[((long long)(b + a.B) * a.HW + pix) * a.Cpad + c] = v;
            }
        }
    }
}

void launch_cfg_step(const LaunchCtx& ctx, const StepArgs& s)
{
    StepDev a;
    a.kind = s.kind; a.pred = s.pred; a.B = s.B; a.C = s.C; a.HW = s.H * s.W; a.cfg = s.cfg; a.guidance = s.guidance;
    a.rescale_factor = s.rescale_factor; a.model_out = s.model_out; a.latents = s.latents; a.noise = s.noise;
    for (int i = 0; i < 12; ++i) a.c[i] = s.c[i];
    a.clip = s.clip; a.dpm_algo = s.dpm_algo;
    for (int i = 0; i < 4; ++i) { a.hist[i] = s.hist[i]; a.lms_c[i] = s.lms_c[i]; }
    a.nhist = s.nhist; a.next_scale = s.next_input_scale; a.next_input = s.next_input; a.Cpad = s.Cpad; a.eps_out = s.eps_out; a.next_f32 = s.next_input_f32; a.x0_out = s.x0_out; a.deriv_out = s.deriv_out;
    SDE_REQUIRE(!((s.kind == STEP_EULER_A) && !s.noise), SDE_ERR_INVALID_ARGUMENT, "cfg_step: Euler-a needs a noise tensor");
    const long long per = (long long)s.B * s.C * s.H * s.W;
    const double bytes = per * ((s.cfg ? 8.0 : 4.0) + 8.0 + (s.noise ? 4.0 : 0.0) + (s.next_input ? (s.cfg ? 4.0 : 2.0) : 0.0) + 4.0 * s.nhist);
    ProfScope ps(ctx, "cfg_scheduler_step", 0, bytes);
    hipLaunchKernelGGL(cfg_step_kernel, dim3((unsigned)std::min<long long>(cdiv64(per, 256), 1024)), dim3(256), 0, ctx.stream, a);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ CFG rescale factor
// impl_txt2img.rs:293-297: ro_pos = std(text half), ro_cfg = std(x_cfg) — SCALAR population std over the whole
// half batch (ndarray `.std(0.)`).  Two-stage reduction of (sum, sumsq) pairs; out = {ro_pos/ro_cfg, m}.
__global__ void rescale_partial_kernel(const float* mo, long long per, float g, float* ws)
{
    float s1 = 0, q1 = 0, s2 = 0, q2 = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        float u = mo[i], t = mo[per + i];
        float x = u + g * (t - u);
        s1 += t; q1 += t * t; s2 += x; q2 += x * x;
    }
    __shared__ float sh[4][256];
    sh[0][threadIdx.x] = s1; sh[1][threadIdx.x] = q1; sh[2][threadIdx.x] = s2; sh[3][threadIdx.x] = q2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int k = 0; k < 4; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; ++k) ws[blockIdx.x * 4 + k] = sh[k][0];
}
__global__ void rescale_final_kernel(const float* ws, int nblocks, long long per, float m, float* out)
{
    if (threadIdx.x != 0) return;
    double s[4] = {0, 0, 0, 0};
    for (int b = 0; b < nblocks; ++b)
        for (int k = 0; k < 4; ++k) s[k] += ws[b * 4 + k];
    double n = (double)per;
    double var_pos = s[1] / n - (s[0] / n) * (s[0] / n), var_cfg = s[3] / n - (s[2] / n) * (s[2] / n);
    out[0] = (float)(sqrt(var_pos > 0 ? var_pos : 0) / sqrt(var_cfg > 0 ? var_cfg : 1e-30));
    out[1] = m;
}
void launch_cfg_rescale_factor(const LaunchCtx& ctx, const float* model_out, int B, int64_t per_half, float guidance, float m,
                               float* ws, float* out)
{
    (void)B;
    ProfScope ps(ctx, "cfg_rescale_std", 0, 8.0 * per_half);
    const int nblocks = 64;
    hipLaunchKernelGGL(rescale_partial_kernel, dim3(nblocks), dim3(256), 0, ctx.stream, model_out, (long long)per_half, guidance, ws);
    hipLaunchKernelGGL(rescale_final_kernel, dim3(1), dim3(64), 0, ctx.stream, ws, nblocks, (long long)per_half, m, out);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ small f32 vector ops
// mode 0: out = x / s ; mode 1: out = a*x + b*y ; mode 2: out = x * s
__global__ void vec_op_kernel(const float* x, const float* y, float* out, long long n, int mode, float a, float b)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v;
        if (mode == 0) v = x[i] / a;
        else if (mode == 1) { float p = a * x[i]; float q = b * y[i]; v = p + q; }
        else v = x[i] * a;
        out[i] = v;
    }
}
void launch_vec_op(const LaunchCtx& ctx, const float* x, const float* y, float* out, int64_t n, int mode, float a, float b)
{
    ProfScope ps(ctx, "vec_op", 0, 8.0 * n);
    hipLaunchKernelGGL(vec_op_kernel, dim3((unsigned)std::min<long long>(cdiv64(n, 256), 2048)), dim3(256), 0, ctx.stream, x, y, out,
                       (long long)n, mode, a, b);
    HIP_CHECK(hipGetLastError());
}
