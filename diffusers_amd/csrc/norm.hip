// norm.hip — GroupNorm(+SiLU) over NHWC, LayerNorm, row softmax.  All HBM-bound (read x once or twice,
// write y once): 16-byte vector accesses, fp32 statistics, wave/LDS reductions.
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------------------------------ GroupNorm
// Pass 1: per (image, pixel-chunk) partial per-channel sums -> per-group partial sums.
//   thread layout: TPP threads across the channel octets of one pixel (each OPT octets), PR pixel rows per block.
// ws layout: part[NB][nchunks][groups][2] | ab[NB][C][2]
#define GN_MAX_OPT 4
__global__ void gn_stats_kernel(const half_t* X0, const half_t* X1, int C0, int C1, int HW, int groups, int TPP, int OPT,
                                int PR, int ppc, int nchunks, float* part)
{
    extern __shared__ float sh[];  // [PR][C] sums | [PR][C] sumsq  (no atomics: results are run-to-run deterministic)
    const int C = C0 + C1;
    const int nb = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x;
    const int cx = tid % TPP, pr = tid / TPP;
    float s[GN_MAX_OPT][8], q[GN_MAX_OPT][8];
#pragma unroll
    for (int o = 0; o < GN_MAX_OPT; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[o][j] = 0.f; q[o][j] = 0.f; }
    const int p_begin = chunk * ppc, p_end = min(HW, p_begin + ppc);
    for (int px = p_begin + pr; px < p_end; px += PR) {
        const long long row = (long long)nb * HW + px;
#pragma unroll
        for (int o = 0; o < GN_MAX_OPT; ++o) {
            if (o < OPT) {
                int c = (cx + o * TPP) * 8;
                const half_t* src = c < C0 ? X0 + row * C0 + c : X1 + row * C1 + (c - C0);
                half8 v = *(const half8*)src;
#pragma unroll
                for (int j = 0; j < 8; ++j) { float f = (float)v[j]; s[o][j] += f; q[o][j] += f * f; }
            }
        }
    }
    float* shs = sh + (size_t)pr * C;
    float* shq = sh + (size_t)PR * C + (size_t)pr * C;
#pragma unroll
    for (int o = 0; o < GN_MAX_OPT; ++o) {
        if (o < OPT) {
            int c = (cx + o * TPP) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { shs[c + j] = s[o][j]; shq[c + j] = q[o][j]; }
        }
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int g = tid; g < groups; g += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < PR; ++r)
            for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += sh[(size_t)r * C + c]; b += sh[(size_t)PR * C + (size_t)r * C + c]; }
        float* dst = part + (((long long)nb * nchunks + chunk) * groups + g) * 2;
        dst[0] = a; dst[1] = b;
    }
}

// Pass 2: reduce partials -> mean / rstd per group -> per-channel scale a = rstd*gamma, shift b = beta - mean*a.
__global__ void gn_finalize_kernel(const float* part, int nchunks, int groups, int C, int HW, float eps, const float* gamma,
                                   const float* beta, float* ab)
{
    const int nb = blockIdx.x;
    __shared__ float smean[64], srstd[64];
    const int cpg = C / groups;
    for (int g = threadIdx.x; g < groups; g += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int ch = 0; ch < nchunks; ++ch) {
            const float* src = part + (((long long)nb * nchunks + ch) * groups + g) * 2;
            a += src[0]; b += src[1];
        }
        float cnt = (float)cpg * (float)HW;
        float mean = a / cnt;
        float var = fmaxf(b / cnt - mean * mean, 0.f);
        smean[g] = mean; srstd[g] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        int g = c / cpg;
        float a = srstd[g] * gamma[c];
        ab[((long long)nb * C + c) * 2] = a;
        ab[((long long)nb * C + c) * 2 + 1] = beta[c] - smean[g] * a;
    }
}

// Pass 3: y = x*a + b (+SiLU); writes the concatenated tensor.
__global__ void gn_apply_kernel(const half_t* X0, const half_t* X1, int C0, int C1, int HW, long long total_oct, const float* ab,
                                int silu, half_t* Y)
{
    const int C = C0 + C1, CX = C >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total_oct; i += (long long)gridDim.x * blockDim.x) {
        long long row = i / CX;
        int c = (int)(i - row * CX) * 8;
        int nb = (int)(row / HW);
        const half_t* src = c < C0 ? X0 + row * C0 + c : X1 + row * C1 + (c - C0);
        half8 v = *(const half8*)src;
        const float* abp = ab + ((long long)nb * C + c) * 2;
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = (float)v[j] * abp[2 * j] + abp[2 * j + 1];
            if (silu) y = y / (1.0f + __expf(-y));
            o[j] = (half_t)y;
        }
        *(half8*)(Y + row * C + c) = o;
    }
}

static void gn_geometry(int C, int HW, int& TPP, int& OPT, int& PR, int& ppc, int& nchunks)
{
    const int CX = C / 8;
    OPT = cdiv(CX, 256);
    while (CX % OPT) ++OPT;
    TPP = CX / OPT;
    PR = std::max(1, 256 / TPP);
    // aim for ~1024 blocks per launch across the batch, at least 64 pixels per chunk
    ppc = std::max(64, cdiv(HW, 64));
    ppc = std::min(ppc, HW);
    nchunks = cdiv(HW, ppc);
}

size_t groupnorm_ws_floats(int NB, int C, int HW)
{
    int TPP, OPT, PR, ppc, nchunks;
    gn_geometry(C, HW, TPP, OPT, PR, ppc, nchunks);
    return (size_t)NB * nchunks * 64 * 2 + (size_t)NB * C * 2;
}

void launch_groupnorm(const LaunchCtx& ctx, const GroupNormArgs& a)
{
    const int C = a.C0 + a.C1;
    SDE_REQUIRE(C % 8 == 0 && a.C0 % 8 == 0 && C % a.groups == 0 && a.groups <= 64, SDE_ERR_INVALID_ARGUMENT,
                "groupnorm: C must be a multiple of 8 and of the group count");
    int TPP, OPT, PR, ppc, nchunks;
    gn_geometry(C, a.HW, TPP, OPT, PR, ppc, nchunks);
    SDE_REQUIRE(OPT <= GN_MAX_OPT, SDE_ERR_INVALID_ARGUMENT, "groupnorm: too many channels");
    float* part = a.ws;
    float* ab = a.ws + (size_t)a.NB * nchunks * 64 * 2;
    const double bytes = (double)a.NB * a.HW * C * 2.0;
    {
        ProfScope ps(ctx, "groupnorm_stats", 0, bytes);
        hipLaunchKernelGGL(gn_stats_kernel, dim3(nchunks, a.NB), dim3(TPP * PR), (size_t)2 * PR * C * sizeof(float), ctx.stream, a.X0, a.X1, a.C0,
                           a.C1, a.HW, a.groups, TPP, OPT, PR, ppc, nchunks, part);
    }
    {
        ProfScope ps(ctx, "groupnorm_finalize", 0, 0);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.NB), dim3(256), 0, ctx.stream, part, nchunks, a.groups, C, a.HW, a.eps, a.gamma,
                           a.beta, ab);
    }
    {
        ProfScope ps(ctx, a.silu ? "groupnorm_silu_apply" : "groupnorm_apply", 0, 2 * bytes);
        long long total = (long long)a.NB * a.HW * (C / 8);
        int blocks = (int)std::min<long long>(cdiv64(total, 256), 256 * 16);
        hipLaunchKernelGGL(gn_apply_kernel, dim3(blocks), dim3(256), 0, ctx.stream, a.X0, a.X1, a.C0, a.C1, a.HW, total, ab,
                           a.silu ? 1 : 0, a.Y);
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; the row lives in registers (<= LN_MAX_OCT octets per lane) for an exact two-pass variance.
#define LN_MAX_OCT 4
__global__ void layernorm_kernel(const half_t* X, half_t* Y, int rows, int C, const float* gamma, const float* beta, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int CX = C >> 3;
    const half_t* x = X + (long long)row * C;
    half8 v[LN_MAX_OCT];
    float sum = 0.f;
#pragma unroll
    for (int o = 0; o < LN_MAX_OCT; ++o) {
        int cx = lane + o * 64;
        if (cx < CX) {
            v[o] = *(const half8*)(x + cx * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (float)v[o][j];
        }
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int o = 0; o < LN_MAX_OCT; ++o) {
        int cx = lane + o * 64;
        if (cx < CX) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { float dlt = (float)v[o][j] - mean; sq += dlt * dlt; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
    half_t* y = Y + (long long)row * C;
#pragma unroll
    for (int o = 0; o < LN_MAX_OCT; ++o) {
        int cx = lane + o * 64;
        if (cx < CX) {
            half8 out;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int c = cx * 8 + j;
                out[j] = (half_t)(((float)v[o][j] - mean) * rstd * gamma[c] + beta[c]);
            }
            *(half8*)(y + cx * 8) = out;
        }
    }
}

void launch_layernorm(const LaunchCtx& ctx, const half_t* X, half_t* Y, int rows, int C, const float* gamma, const float* beta, float eps)
{
    SDE_REQUIRE(C % 8 == 0 && C / 8 <= 64 * LN_MAX_OCT, SDE_ERR_INVALID_ARGUMENT, "layernorm: unsupported width");
    ProfScope ps(ctx, "layernorm", 0, 4.0 * rows * C);
    hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, ctx.stream, X, Y, rows, C, gamma, beta, eps);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ row softmax (VAE attention)
__global__ void softmax_rows_kernel(half_t* X, long long rows, int cols)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    half_t* x = X + row * cols;
    const int CX = cols >> 3;
    float mx = -INFINITY;
    for (int cx = lane; cx < CX; cx += 64) {
        half8 v = *(const half8*)(x + cx * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) mx = fmaxf(mx, (float)v[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int cx = lane; cx < CX; cx += 64) {
        half8 v = *(const half8*)(x + cx * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += __expf((float)v[j] - mx);
    }
    const float inv = 1.0f / wave_sum(sum);
    for (int cx = lane; cx < CX; cx += 64) {
        half8 v = *(const half8*)(x + cx * 8);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(__expf((float)v[j] - mx) * inv);
        *(half8*)(x + cx * 8) = o;
    }
}

void launch_softmax_rows(const LaunchCtx& ctx, half_t* X, int64_t rows, int cols)
{
    SDE_REQUIRE(cols % 8 == 0, SDE_ERR_INVALID_ARGUMENT, "softmax_rows: cols must be a multiple of 8");
    ProfScope ps(ctx, "softmax_rows", 0, 4.0 * rows * cols);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, ctx.stream, X, (long long)rows, cols);
    HIP_CHECK(hipGetLastError());
}
