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
// Two launches, both HBM-bound (x is read twice, y written once):
//   gn_stats_kernel : per (image, pixel-chunk) partial sums per group           -> part[NB][nchunks][groups][2]
//   gn_apply_kernel : each block first folds the partials of its image into per-channel scale/shift in LDS
//                     (a = rstd*gamma, b = beta - mean*a), then streams y = x*a + b (+SiLU) over its rows.
// No atomics anywhere: the reduction order is fixed, results are run-to-run deterministic.
//   thread layout of the stats pass: TPP threads across the channel octets of one pixel (each OPT octets), PR pixel rows.
#define GN_MAX_OPT 4
#define GN_U 8   // pixels per trip: GN_U*OPT 16-byte loads in flight per thread (the pass is latency-bound otherwise)
template <int OPT>
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* X0, const half_t* X1, int C0, int C1, int HW, int groups, int TPP,
                                                       int PR, int ppc, int nchunks, float* part)
{
    extern __shared__ float sh[];  // [PR][C] sums | [PR][C] sumsq
    const int C = C0 + C1;
    const int nb = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x;
    const int cx = tid % TPP, pr = tid / TPP;
    const bool active = pr < PR;
    float s[OPT][8], q[OPT][8];
#pragma unroll
    for (int o = 0; o < OPT; ++o)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[o][j] = 0.f; q[o][j] = 0.f; }
    const int p_begin = chunk * ppc, p_end = min(HW, p_begin + ppc);
    if (active) {
        for (int px = p_begin + pr; px < p_end; px += GN_U * PR) {
            half8 v[GN_U][OPT];
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                const int pp = px + u * PR;
                const long long row = (long long)nb * HW + min(pp, p_end - 1);
#pragma unroll
                for (int o = 0; o < OPT; ++o) {
                    int c = (cx + o * TPP) * 8;
                    const half_t* src = c < C0 ? X0 + row * C0 + c : X1 + row * C1 + (c - C0);
                    v[u][o] = *(const half8*)src;
                }
            }
#pragma unroll
            for (int u = 0; u < GN_U; ++u) {
                if (px + u * PR < p_end) {
#pragma unroll
                    for (int o = 0; o < OPT; ++o) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { float f = (float)v[u][o][j]; s[o][j] += f; q[o][j] += f * f; }
                    }
                }
            }
        }
    }
    if (active) {
        float* shs = sh + (size_t)pr * C;
        float* shq = sh + (size_t)PR * C + (size_t)pr * C;
#pragma unroll
        for (int o = 0; o < OPT; ++o) {
            int c = (cx + o * TPP) * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) { shs[c + j] = s[o][j]; shq[c + j] = q[o][j]; }
        }
    }
    __syncthreads();
    // fold the PR pixel rows: one thread per channel, result in row 0
    for (int c = tid; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < PR; ++r) { a += sh[(size_t)r * C + c]; b += sh[(size_t)PR * C + (size_t)r * C + c]; }
        sh[c] = a; sh[(size_t)PR * C + c] = b;
    }
    __syncthreads();
    // 8 lanes per group (groups <= 32 per pass of 256 threads)
    const int cpg = C / groups;
    for (int g0 = 0; g0 < groups; g0 += 32) {
        const int g = g0 + (tid >> 3), sub = tid & 7;
        float a = 0.f, b = 0.f;
        if (g < groups)
            for (int c = g * cpg + sub; c < (g + 1) * cpg; c += 8) { a += sh[c]; b += sh[(size_t)PR * C + c]; }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (g < groups && sub == 0) {
            float* dst = part + (((long long)nb * nchunks + chunk) * groups + g) * 2;
            dst[0] = a; dst[1] = b;
        }
    }
}

// X0lo / X1lo: low parts of split-fp16 inputs (kernels.h), or null.  The statistics pass reads the high parts only (a 2^-11
// relative perturbation per element moves a group's mean / variance by ~2^-11 / sqrt(count)); the normalised value uses hi + lo.
template <bool LO, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* X0, const half_t* X1, const half_t* X0lo, const half_t* X1lo, int C0, int C1, int HW, int rows_per_block,
                                                       const float* part, int nchunks, int groups, float eps, const float* gamma,
                                                       const float* beta, half_t* Y)
{
    extern __shared__ float sh[];   // ab[C][2] | mean[groups] | rstd[groups]
    const int C = C0 + C1, CX = C >> 3;
    const int nb = blockIdx.y, tid = threadIdx.x;
    float* ab = sh;
    float* smean = sh + 2 * C;
    float* srstd = smean + groups;
    const int cpg = C / groups;
    for (int g0 = 0; g0 < groups; g0 += 32) {
        const int g = g0 + (tid >> 3), sub = tid & 7;
        float a = 0.f, b = 0.f;
        if (g < groups)
            for (int ch = sub; ch < nchunks; ch += 8) {
                const float* src = part + (((long long)nb * nchunks + ch) * groups + g) * 2;
                a += src[0]; b += src[1];
            }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
        if (g < groups && sub == 0) {
            const float cnt = (float)cpg * (float)HW;
            const float mean = a / cnt;
            const float var = fmaxf(b / cnt - mean * mean, 0.f);
            smean[g] = mean; srstd[g] = rsqrtf(var + eps);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += blockDim.x) {
        const int g = c / cpg;
        const float a = srstd[g] * gamma[c];
        ab[2 * c] = a; ab[2 * c + 1] = beta[c] - smean[g] * a;
    }
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_block, r1 = min(HW, r0 + rows_per_block);
    const long long base_row = (long long)nb * HW;
    const int total = (r1 - r0) * CX;
    // 4 octets per trip and two trips in flight: the loads of trip t+1 are issued before trip t is normalised and stored, so a
    // thread's HBM round trips overlap its arithmetic (cvt / fma / exp / rcp: ~12 issue slots per element with SiLU) and its stores
    constexpr int S = 4 * 256;
    struct Trip { half8 v[4], vl[LO ? 4 : 1]; int cc[4]; long long rw[4]; };
    auto load = [&](int i0, Trip& t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * 256, total - 1);
            const int rr = i / CX;
            t.cc[u] = (i - rr * CX) * 8;
            t.rw[u] = base_row + r0 + rr;
            const bool first = t.cc[u] < C0;
            const long long off = first ? t.rw[u] * C0 + t.cc[u] : t.rw[u] * C1 + (t.cc[u] - C0);
            t.v[u] = *(const half8*)((first ? X0 : X1) + off);
            if constexpr (LO) {
                const half_t* lo = first ? X0lo : X1lo;
                if (lo) t.vl[u] = *(const half8*)(lo + off);
                else
#pragma unroll
                    for (int j = 0; j < 8; ++j) t.vl[u][j] = (half_t)0.f;
            }
        }
    };
    auto emit = [&](int i0, const Trip& t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * 256 < total) {
                const float4* abp = (const float4*)(ab + 2 * t.cc[u]);
                half8 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 q = abp[j];
                    float x0 = (float)t.v[u][2 * j], x1 = (float)t.v[u][2 * j + 1];
                    if constexpr (LO) { x0 += (float)t.vl[u][2 * j]; x1 += (float)t.vl[u][2 * j + 1]; }
                    float y0 = x0 * q.x + q.y, y1 = x1 * q.z + q.w;
                    if constexpr (SILU) { y0 = y0 * __builtin_amdgcn_rcpf(1.0f + __expf(-y0)); y1 = y1 * __builtin_amdgcn_rcpf(1.0f + __expf(-y1)); }   // v_rcp_f32 (1 ulp), not the 11-instruction IEEE division
                    o[2 * j] = (half_t)y0; o[2 * j + 1] = (half_t)y1;
                }
                *(half8*)(Y + t.rw[u] * C + t.cc[u]) = o;
            }
        }
    };
    Trip ta, tb;
    int i0 = tid;
    if (i0 < total) load(i0, ta);
    while (i0 < total) {
        if (i0 + S < total) load(i0 + S, tb);
        emit(i0, ta);
        i0 += S;
        if (i0 >= total) break;
        if (i0 + S < total) load(i0 + S, ta);
        emit(i0, tb);
        i0 += S;
    }
}

// Small tensors (deep UNet levels): one block per (image, group) holds the whole group in registers: one launch, x read
// once, no workspace.  The two-pass path above costs two ~13 us launch/latency floors there for a few MB of data.
// Needs cpg % 4 == 0 (8-byte chunks never straddle a group or the concat boundary) and HW*cpg/4 <= 256*GN_SMALL_MAX.
#define GN_SMALL_MAX 24
typedef _Float16 half4n __attribute__((ext_vector_type(4)));
template <bool LO>
__global__ __launch_bounds__(256) void gn_small_kernel(const half_t* X0, const half_t* X1, const half_t* X0lo, const half_t* X1lo, int C0, int C1, int HW, int groups, float eps,
                                                       const float* gamma, const float* beta, int silu, half_t* Y)
{
    __shared__ float red[2][4];
    const int C = C0 + C1, cpg = C / groups, cq = cpg >> 2;   // 8-byte chunks per pixel in this group
    const int g = blockIdx.x, nb = blockIdx.y, tid = threadIdx.x;
    const int total = HW * cq;
    const int c_base = g * cpg;
    half4n v[GN_SMALL_MAX], vl[LO ? GN_SMALL_MAX : 1];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int u = 0; u < GN_SMALL_MAX; ++u) {
        const int i = tid + u * 256;
        if (i < total) {
            const int px = i / cq, c = c_base + (i - px * cq) * 4;
            const long long row = (long long)nb * HW + px;
            const bool first = c < C0;
            const long long off = first ? row * C0 + c : row * C1 + (c - C0);
            v[u] = *(const half4n*)((first ? X0 : X1) + off);
            if (LO) {
                const half_t* lo = first ? X0lo : X1lo;
                if (lo) vl[u] = *(const half4n*)(lo + off);
                else vl[u] = half4n{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
            }
        }
    }
#pragma unroll
    for (int u = 0; u < GN_SMALL_MAX; ++u) {
        if (tid + u * 256 < total) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float f = LO ? (float)v[u][j] + (float)vl[u][j] : (float)v[u][j]; s += f; q += f * f; }
        }
    }
    s = wave_sum(s); q = wave_sum(q);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = q; }
    __syncthreads();
    s = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    const float cnt = (float)cpg * (float)HW;
    const float mean = s / cnt;
    const float rstd = rsqrtf(fmaxf(q / cnt - mean * mean, 0.f) + eps);
#pragma unroll
    for (int u = 0; u < GN_SMALL_MAX; ++u) {
        const int i = tid + u * 256;
        if (i < total) {
            const int px = i / cq, c = c_base + (i - px * cq) * 4;
            const float4 ga = *(const float4*)(gamma + c), be = *(const float4*)(beta + c);
            const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
            half4n o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = rstd * gg[j];
                const float xv = LO ? (float)v[u][j] + (float)vl[u][j] : (float)v[u][j];
                float y = xv * a + (bb[j] - mean * a);   // same scale/shift form as the two-pass path
                if (silu) y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
                o[j] = (half_t)y;
            }
            *(half4n*)(Y + ((long long)nb * HW + px) * C + c) = o;
        }
    }
}

static void gn_geometry(int C, int HW, int& TPP, int& OPT, int& PR, int& ppc, int& nchunks)
{
    const int CX = C / 8;
    OPT = cdiv(CX, 256);
    while (CX % OPT) ++OPT;
    TPP = CX / OPT;
    PR = std::max(1, 256 / TPP);
    // <= 64 pixel chunks per image, at least 64 pixels each
    ppc = std::max(64, cdiv(HW, 64));
    ppc = std::min(ppc, HW);
    nchunks = cdiv(HW, ppc);
}

size_t groupnorm_ws_floats(int NB, int C, int HW)
{
    int TPP, OPT, PR, ppc, nchunks;
    gn_geometry(C, HW, TPP, OPT, PR, ppc, nchunks);
    // fp32 engine (f32path.hip): [NB][<= 64 slices][<= 64 groups][2] doubles
    return std::max((size_t)NB * nchunks * 64 * 2, (size_t)NB * 64 * 64 * 4);
}

void launch_groupnorm(const LaunchCtx& ctx, const GroupNormArgs& a)
{
    const int C = a.C0 + a.C1;
    if (ctx.f32) { f32_launch_groupnorm(ctx, a); return; }
    SDE_REQUIRE(C % 8 == 0 && a.C0 % 8 == 0 && C % a.groups == 0 && a.groups <= 64, SDE_ERR_INVALID_ARGUMENT,
                "groupnorm: C must be a multiple of 8 and of the group count");
    int TPP, OPT, PR, ppc, nchunks;
    gn_geometry(C, a.HW, TPP, OPT, PR, ppc, nchunks);
    SDE_REQUIRE(OPT <= GN_MAX_OPT, SDE_ERR_INVALID_ARGUMENT, "groupnorm: too many channels");
    float* part = a.ws;
    const double bytes = (double)a.NB * a.HW * C * 2.0;
    {
        const int cpg = C / a.groups;
        if (cpg % 4 == 0 && a.C0 % 4 == 0 && (long long)a.HW * (cpg / 4) <= 256LL * 16 &&   // beyond ~16 chunks per thread the two-pass path is faster (measured)
           
            ((reinterpret_cast<uintptr_t>(a.gamma) | reinterpret_cast<uintptr_t>(a.beta)) & 15) == 0) {
            char nm[48];
            static const bool shp = getenv("SDE_PROF_SHAPES") && atoi(getenv("SDE_PROF_SHAPES"));
            if (shp) snprintf(nm, sizeof(nm), "gn1 HW%d C%d", a.HW, C); else snprintf(nm, sizeof(nm), "%s", a.silu ? "groupnorm_silu_fused_small" : "groupnorm_fused_small");
            ProfScope ps(ctx, nm, 0, 2 * bytes);
            if (a.X0lo || a.X1lo)
                hipLaunchKernelGGL(gn_small_kernel<true>, dim3(a.groups, a.NB), dim3(256), 0, ctx.stream, a.X0, a.X1, a.X0lo, a.X1lo, a.C0, a.C1, a.HW,
                                   a.groups, a.eps, a.gamma, a.beta, a.silu ? 1 : 0, a.Y);
            else
                hipLaunchKernelGGL(gn_small_kernel<false>, dim3(a.groups, a.NB), dim3(256), 0, ctx.stream, a.X0, a.X1, a.X0lo, a.X1lo, a.C0, a.C1, a.HW,
                                   a.groups, a.eps, a.gamma, a.beta, a.silu ? 1 : 0, a.Y);
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    static const bool shapes = getenv("SDE_PROF_SHAPES") && atoi(getenv("SDE_PROF_SHAPES"));
    char nm_s[48] = "groupnorm_stats", nm_a[48];
    snprintf(nm_a, sizeof(nm_a), "%s", a.silu ? "groupnorm_silu_apply" : "groupnorm_apply");
    if (shapes) { snprintf(nm_s, sizeof(nm_s), "gnS HW%d C%d", a.HW, C); snprintf(nm_a, sizeof(nm_a), "gnA HW%d C%d", a.HW, C); }
    {
        ProfScope ps(ctx, nm_s, 0, bytes);
        const dim3 grid(nchunks, a.NB), block(256);
        const size_t lds = (size_t)2 * PR * C * sizeof(float);
#define GN_STATS(O) hipLaunchKernelGGL(gn_stats_kernel<O>, grid, block, lds, ctx.stream, a.X0, a.X1, a.C0, a.C1, a.HW, a.groups, TPP, PR, ppc, nchunks, part)
        if (OPT == 1) GN_STATS(1); else if (OPT == 2) GN_STATS(2); else if (OPT == 3) GN_STATS(3); else GN_STATS(4);
#undef GN_STATS
    }
    {
        ProfScope ps(ctx, nm_a, 0, 2 * bytes + (double)a.NB * a.HW * 2.0 * ((a.X0lo ? a.C0 : 0) + (a.X1lo ? a.C1 : 0)));
        // ~1024 blocks across the batch, at least 16 pixel rows each
        int bpi = std::max(1, std::min(cdiv(1024, a.NB), cdiv(a.HW, 16)));
        const int rpb = cdiv(a.HW, bpi);
        bpi = cdiv(a.HW, rpb);
#define GN_APPLY(LO_, SILU_) hipLaunchKernelGGL((gn_apply_kernel<LO_, SILU_>), dim3(bpi, a.NB), dim3(256), (size_t)(2 * C + 2 * a.groups) * sizeof(float), ctx.stream, \
                                                a.X0, a.X1, a.X0lo, a.X1lo, a.C0, a.C1, a.HW, rpb, part, nchunks, a.groups, a.eps, a.gamma, a.beta, a.Y)
        const bool lo = a.X0lo || a.X1lo;
        if (lo) { if (a.silu) GN_APPLY(true, true); else GN_APPLY(true, false); }
        else    { if (a.silu) GN_APPLY(false, true); else GN_APPLY(false, false); }
#undef GN_APPLY
    }
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ LayerNorm
// one wave per LN_ROWS rows (4 for long tensors, 1 when there are too few rows to fill the chip otherwise); the rows live in registers (<= LN_MAX_OCT octets per lane and row) for an exact two-pass
// variance.  All loads of the wave's rows are issued before the first reduction (memory-level parallelism).
#define LN_MAX_OCT 4
template <int NOCT, int LN_ROWS>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* X, half_t* Y, int rows, int C, const float* gamma, const float* beta, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * LN_ROWS;
    if (row0 >= rows) return;
    const int CX = C >> 3;
    half8 v[LN_ROWS][NOCT];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const half_t* x = X + (long long)min(row0 + r, rows - 1) * C;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            const int cx = lane + o * 64;
            if (cx < CX) v[r][o] = *(const half8*)(x + cx * 8);
        }
    }
    float g[NOCT][8], bta[NOCT][8];
#pragma unroll
    for (int o = 0; o < NOCT; ++o) {
        const int cx = lane + o * 64;
        if (cx < CX) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { g[o][j] = gamma[cx * 8 + j]; bta[o][j] = beta[cx * 8 + j]; }
        }
    }
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            if (lane + o * 64 < CX) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += (float)v[r][o][j];
            }
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            if (lane + o * 64 < CX) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float dlt = (float)v[r][o][j] - mean; sq += dlt * dlt; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        if (row0 + r < rows) {
            half_t* y = Y + (long long)(row0 + r) * C;
#pragma unroll
            for (int o = 0; o < NOCT; ++o) {
                const int cx = lane + o * 64;
                if (cx < CX) {
                    half8 out;
#pragma unroll
                    for (int j = 0; j < 8; ++j) out[j] = (half_t)(((float)v[r][o][j] - mean) * rstd * g[o][j] + bta[o][j]);
                    *(half8*)(y + cx * 8) = out;
                }
            }
        }
    }
}

// Lane-dense LayerNorm for the UNet's widths (C = 320 / 640 / 1280 = 40 L octets... i.e. L = 8 / 16 / 32 lanes per row, five 16-byte chunks
// per lane): the kernels above put one chunk on each lane, which at C = 320 leaves 24 of 64 lanes idle in every memory instruction.
// Here a wave normalises 64 / L rows at a time with every lane busy (lane l of a row's group holds chunks l, l + L, .. l + 4L, so the
// group's loads of one instruction are L x 16 contiguous bytes), keeps its gamma / beta slice in registers and walks a strided list
// of row groups.  y = LN(A + B + Cc) with the sum formed in fp32 in the given order (B, Cc optional); exact two-pass variance.
template <int L, int NIN>
__global__ __launch_bounds__(256) void layernorm_lanes_kernel(const half_t* A, const half_t* B, const half_t* Cc, half_t* Y, int rows, int C,
                                                              const float* gamma, const float* beta, float eps)
{
    constexpr int RPW = 64 / L;
    const int lane = threadIdx.x & 63, l = lane % L, sub = lane / L;
    float g[5][8], bt[5][8];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const float4 g0 = *(const float4*)(gamma + (l + L * i) * 8), g1 = *(const float4*)(gamma + (l + L * i) * 8 + 4);
        const float4 b0 = *(const float4*)(beta + (l + L * i) * 8), b1 = *(const float4*)(beta + (l + L * i) * 8 + 4);
        g[i][0] = g0.x; g[i][1] = g0.y; g[i][2] = g0.z; g[i][3] = g0.w; g[i][4] = g1.x; g[i][5] = g1.y; g[i][6] = g1.z; g[i][7] = g1.w;
        bt[i][0] = b0.x; bt[i][1] = b0.y; bt[i][2] = b0.z; bt[i][3] = b0.w; bt[i][4] = b1.x; bt[i][5] = b1.y; bt[i][6] = b1.z; bt[i][7] = b1.w;
    }
    const int ngroups = (rows + RPW - 1) / RPW, stride = gridDim.x * (blockDim.x >> 6);
    const float invC = 1.0f / (float)C;
    for (int grp = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); grp < ngroups; grp += stride) {
        const int row = grp * RPW + sub;
        const long long off = (long long)min(row, rows - 1) * C + l * 8;
        half8 a[5], b[NIN > 1 ? 5 : 1], c[NIN > 2 ? 5 : 1];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            a[i] = *(const half8*)(A + off + L * 8 * i);
            if constexpr (NIN > 1) b[i] = *(const half8*)(B + off + L * 8 * i);
            if constexpr (NIN > 2) c[i] = *(const half8*)(Cc + off + L * 8 * i);
        }
        float v[5][8];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float x = (float)a[i][j];
                if constexpr (NIN > 1) x += (float)b[i][j];
                if constexpr (NIN > 2) x += (float)c[i][j];
                v[i][j] = x;
                sum += x;
            }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        const float mean = sum * invC;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq = fmaf(d, d, sq); }
#pragma unroll
        for (int o = L / 2; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
        const float rstd = rsqrtf(sq * invC + eps);
        if (row < rows) {
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                half8 out;
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] = (half_t)((v[i][j] - mean) * rstd * g[i][j] + bt[i][j]);
                *(half8*)(Y + off + L * 8 * i) = out;
            }
        }
    }
}
// true if launched
static bool launch_layernorm_lanes(const LaunchCtx& ctx, const half_t* A, const half_t* B, const half_t* Cc, half_t* Y, int rows, int C,
                                   const float* gamma, const float* beta, float eps)
{
    static const int env = getenv("SDE_LN_LANES") ? atoi(getenv("SDE_LN_LANES")) : 1;   // tuning hook: 0 = the one-chunk-per-lane kernels
    const int L = C / 40;
    if (!env || C % 40 || (L != 8 && L != 16 && L != 32) || rows < 1024 || (Cc && !B)) return false;
    const int nin = Cc ? 3 : (B ? 2 : 1), rpw = 64 / L;
    const int grid = std::max(1, std::min(cdiv(rows, rpw * 4 * 2), 2048));      // two or more row groups per wave
#define LNL(L_, N_) hipLaunchKernelGGL((layernorm_lanes_kernel<L_, N_>), dim3(grid), dim3(256), 0, ctx.stream, A, B, Cc, Y, rows, C, gamma, beta, eps)
#define LNL_N(L_) do { if (nin == 1) LNL(L_, 1); else if (nin == 2) LNL(L_, 2); else LNL(L_, 3); } while (0)
    if (L == 8) LNL_N(8); else if (L == 16) LNL_N(16); else LNL_N(32);
#undef LNL_N
#undef LNL
    return true;
}

void launch_layernorm(const LaunchCtx& ctx, const half_t* X, half_t* Y, int rows, int C, const float* gamma, const float* beta, float eps)
{
    if (ctx.f32) { f32_launch_layernorm_sum(ctx, nullptr, X, nullptr, nullptr, Y, rows, C, gamma, beta, eps); return; }
    SDE_REQUIRE(C % 8 == 0 && C / 8 <= 64 * LN_MAX_OCT, SDE_ERR_INVALID_ARGUMENT, "layernorm: unsupported width");
    char nm_l[48] = "layernorm";
    if (getenv("SDE_PROF_SHAPES") && atoi(getenv("SDE_PROF_SHAPES"))) snprintf(nm_l, sizeof(nm_l), "ln M%d C%d", rows, C);
    ProfScope ps(ctx, nm_l, 0, 4.0 * rows * C);
    if (launch_layernorm_lanes(ctx, X, nullptr, nullptr, Y, rows, C, gamma, beta, eps)) { HIP_CHECK(hipGetLastError()); return; }
    const int noct = cdiv(C / 8, 64);
    const dim3 block(256);
#define LN_LAUNCH(O, R) hipLaunchKernelGGL((layernorm_kernel<O, R>), dim3(cdiv(rows, 4 * R)), block, 0, ctx.stream, X, Y, rows, C, gamma, beta, eps)
    if (rows >= 32768) {
        if (noct == 1) LN_LAUNCH(1, 4); else if (noct == 2) LN_LAUNCH(2, 4); else if (noct == 3) LN_LAUNCH(3, 4); else LN_LAUNCH(4, 4);
    } else {
        if (noct == 1) LN_LAUNCH(1, 1); else if (noct == 2) LN_LAUNCH(2, 1); else if (noct == 3) LN_LAUNCH(3, 1); else LN_LAUNCH(4, 1);
    }
#undef LN_LAUNCH
    HIP_CHECK(hipGetLastError());
}


// LayerNorm over a SUM kept in fp32: x = S32 + A + B (each optional but one), y = LN(x) in fp16, optionally x written back as
// fp32.  This is how the transformer blocks carry their residual stream (model.cpp): the branch outputs (attention, feed-forward)
// stay fp16 GEMM outputs, the running sum is never rounded to fp16 between sub-layers (the rounding of the
// residual stream at every residual add was the largest single contribution to the engine-vs-fp32 error).
template <int NOCT, int LN_ROWS>
__global__ __launch_bounds__(256) void layernorm_sum_kernel(const float* S32, const half_t* A, const half_t* B, const half_t* Cc, float* S32_out, half_t* Y,
                                                            int rows, int C, const float* gamma, const float* beta, float eps)
{
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * LN_ROWS;
    if (row0 >= rows) return;
    const int CX = C >> 3;
    float v[LN_ROWS][NOCT][8];
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        const long long off = (long long)min(row0 + r, rows - 1) * C;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            const int cx = lane + o * 64;
            if (cx < CX) {
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
                if (S32) { s0 = *(const float4*)(S32 + off + cx * 8); s1 = *(const float4*)(S32 + off + cx * 8 + 4); }
                v[r][o][0] = s0.x; v[r][o][1] = s0.y; v[r][o][2] = s0.z; v[r][o][3] = s0.w;
                v[r][o][4] = s1.x; v[r][o][5] = s1.y; v[r][o][6] = s1.z; v[r][o][7] = s1.w;
                if (A) {
                    half8 a = *(const half8*)(A + off + cx * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[r][o][j] += (float)a[j];
                }
                if (B) {
                    half8 b = *(const half8*)(B + off + cx * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[r][o][j] += (float)b[j];
                }
                if (Cc) {
                    half8 c = *(const half8*)(Cc + off + cx * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[r][o][j] += (float)c[j];
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < LN_ROWS; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            if (lane + o * 64 < CX) {
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += v[r][o][j];
            }
        }
        const float mean = wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int o = 0; o < NOCT; ++o) {
            if (lane + o * 64 < CX) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { float dlt = v[r][o][j] - mean; sq += dlt * dlt; }
            }
        }
        const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
        if (row0 + r < rows) {
            const long long off = (long long)(row0 + r) * C;
#pragma unroll
            for (int o = 0; o < NOCT; ++o) {
                const int cx = lane + o * 64;
                if (cx < CX) {
                    if (S32_out) {
                        *(float4*)(S32_out + off + cx * 8) = make_float4(v[r][o][0], v[r][o][1], v[r][o][2], v[r][o][3]);
                        *(float4*)(S32_out + off + cx * 8 + 4) = make_float4(v[r][o][4], v[r][o][5], v[r][o][6], v[r][o][7]);
                    }
                    half8 out;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        out[j] = (half_t)((v[r][o][j] - mean) * rstd * gamma[cx * 8 + j] + beta[cx * 8 + j]);
                    *(half8*)(Y + off + cx * 8) = out;
                }
            }
        }
    }
}

static void launch_layernorm_sum_impl(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, const half_t* Cc, float* S32_out, half_t* Y,
                                      int rows, int C, const float* gamma, const float* beta, float eps);
void launch_layernorm_sum(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, float* S32_out, half_t* Y, int rows, int C,
                          const float* gamma, const float* beta, float eps)
{
    SDE_REQUIRE(S32 || A || B, SDE_ERR_INVALID_ARGUMENT, "layernorm_sum: no input");
    if (ctx.f32) { f32_launch_layernorm_sum(ctx, S32, A, B, S32_out, Y, rows, C, gamma, beta, eps); return; }
    launch_layernorm_sum_impl(ctx, S32, A, B, nullptr, S32_out, Y, rows, C, gamma, beta, eps);
}
void launch_layernorm_sum3(const LaunchCtx& ctx, const half_t* A, const half_t* B, const half_t* Cc, half_t* Y, int rows, int C, const float* gamma,
                           const float* beta, float eps)
{
    SDE_REQUIRE(!ctx.f32 && A && B, SDE_ERR_INVALID_ARGUMENT, "layernorm_sum3: fp16 engine, at least two terms");
    launch_layernorm_sum_impl(ctx, nullptr, A, B, Cc, nullptr, Y, rows, C, gamma, beta, eps);
}
static void launch_layernorm_sum_impl(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, const half_t* Cc, float* S32_out, half_t* Y,
                                      int rows, int C, const float* gamma, const float* beta, float eps)
{
    SDE_REQUIRE(C % 8 == 0 && C / 8 <= 64 * LN_MAX_OCT, SDE_ERR_INVALID_ARGUMENT, "layernorm: unsupported width");
    ProfScope ps(ctx, "layernorm_sum", 0, (double)rows * C * ((S32 ? 4.0 : 0.0) + (A ? 2.0 : 0.0) + (B ? 2.0 : 0.0) + (Cc ? 2.0 : 0.0) + (S32_out ? 4.0 : 0.0) + 2.0));
    if (!S32 && !S32_out && A && launch_layernorm_lanes(ctx, A, B, Cc, Y, rows, C, gamma, beta, eps)) { HIP_CHECK(hipGetLastError()); return; }
    const int noct = cdiv(C / 8, 64);
    const dim3 block(256);
#define LN_LAUNCH(O, R) hipLaunchKernelGGL((layernorm_sum_kernel<O, R>), dim3(cdiv(rows, 4 * R)), block, 0, ctx.stream, S32, A, B, Cc, S32_out, Y, rows, C, gamma, beta, eps)
    if (rows >= 32768 && noct <= 2) {
        if (noct == 1) LN_LAUNCH(1, 4); else LN_LAUNCH(2, 4);
    } else if (rows >= 8192) {
        if (noct == 1) LN_LAUNCH(1, 2); else if (noct == 2) LN_LAUNCH(2, 2); else if (noct == 3) LN_LAUNCH(3, 2); else LN_LAUNCH(4, 2);
    } else {
        if (noct == 1) LN_LAUNCH(1, 1); else if (noct == 2) LN_LAUNCH(2, 1); else if (noct == 3) LN_LAUNCH(3, 1); else LN_LAUNCH(4, 1);
    }
#undef LN_LAUNCH
    HIP_CHECK(hipGetLastError());
}

// y = half(S32 + A + B): the one rounding of a transformer block's residual stream, before proj_out reads it as an MFMA operand
__global__ void sum_round_kernel(const float* S32, const half_t* A, const half_t* B, half_t* Y, half_t* Ylo, long long n8)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const float4 s0 = *(const float4*)(S32 + i * 8), s1 = *(const float4*)(S32 + i * 8 + 4);
        const half8 a = *(const half8*)(A + i * 8), b = *(const half8*)(B + i * 8);
        const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        half8 o, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = s[j] + (float)a[j] + (float)b[j];
            o[j] = (half_t)x;
            l[j] = (half_t)(x - (float)o[j]);
        }
        *(half8*)(Y + i * 8) = o;
        if (Ylo) *(half8*)(Ylo + i * 8) = l;
    }
}
// y = half(((A + B) + C) + D) in fp32: the same block-final sum from the four fp16 branch outputs themselves (no fp32 tensor in between)
__global__ void sum_round4_kernel(const half_t* A, const half_t* B, const half_t* Cc, const half_t* Dd, half_t* Y, half_t* Ylo, long long n8)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const half8 a = *(const half8*)(A + i * 8), b = *(const half8*)(B + i * 8), c = *(const half8*)(Cc + i * 8), d = *(const half8*)(Dd + i * 8);
        half8 o, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (((float)a[j] + (float)b[j]) + (float)c[j]) + (float)d[j];
            o[j] = (half_t)x;
            l[j] = (half_t)(x - (float)o[j]);
        }
        *(half8*)(Y + i * 8) = o;
        if (Ylo) *(half8*)(Ylo + i * 8) = l;
    }
}
void launch_sum_round4(const LaunchCtx& ctx, const half_t* A, const half_t* B, const half_t* Cc, const half_t* Dd, half_t* Y, half_t* Ylo, int64_t n)
{
    SDE_REQUIRE(!ctx.f32 && n % 8 == 0 && A && B && Cc && Dd, SDE_ERR_INVALID_ARGUMENT, "sum_round4: bad arguments");
    ProfScope ps(ctx, "residual_sum_round", 0, (Ylo ? 12.0 : 10.0) * n);
    hipLaunchKernelGGL(sum_round4_kernel, dim3((unsigned)std::min<long long>(cdiv64(n / 8, 256), 8192)), dim3(256), 0, ctx.stream, A, B, Cc, Dd, Y, Ylo,
                       (long long)(n / 8));
    HIP_CHECK(hipGetLastError());
}
void launch_sum_round(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, half_t* Y, half_t* Ylo, int64_t n)
{
    if (ctx.f32) { SDE_REQUIRE(!Ylo && S32 && A && B, SDE_ERR_INVALID_ARGUMENT, "sum_round: bad arguments"); f32_launch_sum_round(ctx, S32, A, B, Y, n); return; }
    SDE_REQUIRE(n % 8 == 0 && S32 && A && B, SDE_ERR_INVALID_ARGUMENT, "sum_round: bad arguments");
    ProfScope ps(ctx, "residual_sum_round", 0, (Ylo ? 12.0 : 10.0) * n);
    hipLaunchKernelGGL(sum_round_kernel, dim3((unsigned)std::min<long long>(cdiv64(n / 8, 256), 8192)), dim3(256), 0, ctx.stream, S32, A, B, Y,
                       Ylo, (long long)(n / 8));
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------ row softmax (VAE attention)
__global__ void softmax_rows_kernel(half_t* X, long long rows, int cols, int pitch)
{
    // rows of `pitch` elements (a multiple of 8); the first `cols` are scores, the rest is padding that is written as zeros
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    half_t* x = X + row * pitch;
    const int CX = pitch >> 3;
    float mx = -INFINITY;
    for (int cx = lane; cx < CX; cx += 64) {
        half8 v = *(const half8*)(x + cx * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (cx * 8 + j < cols) mx = fmaxf(mx, (float)v[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int cx = lane; cx < CX; cx += 64) {
        half8 v = *(const half8*)(x + cx * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (cx * 8 + j < cols) sum += __expf((float)v[j] - mx);
    }
    const float inv = 1.0f / wave_sum(sum);
    for (int cx = lane; cx < CX; cx += 64) {
        half8 v = *(const half8*)(x + cx * 8);
        half8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = cx * 8 + j < cols ? (half_t)(__expf((float)v[j] - mx) * inv) : (half_t)0.f;
        *(half8*)(x + cx * 8) = o;
    }
}

void launch_softmax_rows(const LaunchCtx& ctx, half_t* X, int64_t rows, int cols, int pitch)
{
    SDE_REQUIRE(pitch % 8 == 0 && cols > 0 && cols <= pitch, SDE_ERR_INVALID_ARGUMENT, "softmax_rows: pitch must be a multiple of 8 and >= cols");
    ProfScope ps(ctx, "softmax_rows", 0, 4.0 * rows * cols);
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, ctx.stream, X, (long long)rows, cols, pitch);
    HIP_CHECK(hipGetLastError());
}
