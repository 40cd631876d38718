// f32path.hip — the fp32 engine (sde_engine_opts.dtype = SDE_F32): every activation and weight is a float, every kernel below
// computes in fp32.  It exists for the reference's own precision (its tensors and ORT graphs are f32: impl_txt2img.rs:262,283;
// sd2pyke.py:86,120-125) and for the north-star's 1e-4 parity bar; it is a correctness path, not the benchmarked one: GEMM /
// implicit-GEMM conv on v_mfma_f32_16x16x4_f32 tiles (fp32 operands; the plain LDS-tiled FMA kernel takes the shapes the tiles cannot),
// two-pass attention, three-pass GroupNorm.  Same contracts as the fp16 launchers in kernels.h: the launchers there branch here when
// LaunchCtx::f32 is set, with every `half_t*` argument actually addressing floats.
#include "gemm_common.h"

namespace {

__device__ __forceinline__ float wsum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------ GEMM / implicit-GEMM conv
struct GemmF {
    const float* A0; const float* A1;
    int C0, C1, lda0, lda1;
    int M, N, K, taps, stride, ups, Hi, Wi, Ho, Wo, Hs, Ws, coff;
    const float* W; int Npad;
    const float* bias; const float* R; int ldr;
    float* D; float* Df; int ldd; int epi; float alpha;
    long long sA, sW, sD, sR;
};

// element (m, k) of the implicit im2col matrix; k runs in the engine's weight order ((c / ck) * taps + tap) * ck + c % ck (kernels.h)
__device__ __forceinline__ float a_at(const GemmF& p, const float* A0, const float* A1, int m, int k)
{
    if (m >= p.M || k >= p.K) return 0.f;
    const int Ctot = p.C0 + p.C1;
    int tap = 0, c = k;
    long long pix = m;
    if (p.taps == 9) {
        const int ck = (Ctot % 64 == 0) ? 64 : Ctot;
        const int chunk = k / (9 * ck), rem = k - chunk * 9 * ck;
        tap = rem / ck;
        c = chunk * ck + (rem - tap * ck);
        const int hw = p.Ho * p.Wo;
        const int nb = m / hw, r = m - nb * hw, oy = r / p.Wo, ox = r - oy * p.Wo;
        const int iy = oy * p.stride + p.coff - 1 + tap / 3, ix = ox * p.stride + p.coff - 1 + tap % 3;
        if ((unsigned)iy >= (unsigned)p.Hi || (unsigned)ix >= (unsigned)p.Wi) return 0.f;
        pix = (long long)nb * p.Hs * p.Ws + (long long)(iy >> p.ups) * p.Ws + (ix >> p.ups);
    }
    return c < p.C0 ? A0[pix * p.lda0 + c] : A1[pix * p.lda1 + (c - p.C0)];
}

constexpr int BM = 64, BN = 64, BK = 16;
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmF p)
{
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int z = blockIdx.z;
    const float* A0 = p.A0 + z * p.sA;
    const float* A1 = p.A1 ? p.A1 + z * p.sA : nullptr;
    const float* Wp = p.W + z * p.sW;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < p.K; k0 += BK) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int id = tid + e * 256, ml = id >> 4, kl = id & 15;      // k fastest across lanes: coalesced in both operands
            As[kl][ml] = a_at(p, A0, A1, m0 + ml, k0 + kl);
            const int n = n0 + ml;
            Bs[kl][ml] = (n < p.Npad && k0 + kl < p.K) ? Wp[(long long)n * p.K + k0 + kl] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    const long long zD = z * p.sD, zR = z * p.sR;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float x = acc[i][j] * p.alpha + (p.bias ? p.bias[n] : 0.f);
            if (p.epi == EPI_STD) {
                if (p.R) x += p.R[zR + (long long)m * p.ldr + n];
                p.D[zD + (long long)m * p.ldd + n] = x;
            } else if (p.epi == EPI_NCHW_F32) {
                const int hw = p.Ho * p.Wo, nb = m / hw, pix = m - nb * hw;
                p.Df[zD + ((long long)nb * p.N + n) * hw + pix] = x;
            } else if (p.epi == EPI_IMAGE) {
                x = x / 2.0f + 0.5f;
                p.Df[zD + (long long)m * p.N + n] = fminf(fmaxf(x, 0.0f), 1.0f);
            } else {   // EPI_TRANS
                p.D[zD + (long long)n * p.ldd + m] = x;
            }
        }
    }
}

// MFMA variant: v_mfma_f32_16x16x4_f32 (f32 operands, f32 accumulation: each product and sum is the same IEEE operation as the fmaf chain
// of the plain kernel above, in another order).  256 threads = 2 x 2 waves, block tile BMT x BNT x 16, two LDS stages, the next slab's
// global loads in flight over the current slab's MFMAs, one barrier per slab.
//   * Both operands sit in LDS as [row][16 k + 4 pad] floats.  Lane (i = lane % 16, g = lane / 16) reads ONE b128 per 16-row tile and
//     slab: k = 4g .. 4g+3 of row i.  MFMA number s of the slab takes component s from every lane, i.e. the hardware's "k = g" slot
//     carries k = 4g + s: A and B use the same permutation, so the four MFMAs together cover the slab's 16 k once each.
//   * A 16-float slab never straddles a tap, a 64-channel chunk or the concat boundary (launcher checks C0 % 16, ck % 16): tap and
//     source select are wave-uniform per slab, the per-row pixel decomposition is done once per thread.
template <int BMT, int BNT>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmF p)
{
    constexpr int LDT = 20;                            // 80-byte rows: b128 reads of 16 consecutive rows tile all 64 banks
    constexpr int EA = BMT / 64, EB = BNT / 64;        // float4 loads per thread and slab
    constexpr int TM = BMT / 32, TN = BNT / 32;        // 16 x 16 tiles per wave
    __shared__ __align__(16) float As[2][BMT * LDT];
    __shared__ __align__(16) float Bs[2][BNT * LDT];
    const int z = blockIdx.z;
    const float* A0 = p.A0 + z * p.sA;
    const float* A1 = p.A1 ? p.A1 + z * p.sA : nullptr;
    const float* Wp = p.W + z * p.sW;
    const int m0 = blockIdx.x * BMT, n0 = blockIdx.y * BNT;
    const int tid = threadIdx.x, lrow = tid >> 2, kq = (tid & 3) * 4;
    const int Ctot = p.C0 + p.C1, ck = (Ctot % 64 == 0) ? 64 : Ctot;
    const int hw = p.Ho * p.Wo;

    int iy0[EA], ix0[EA];
    long long base[EA];                                // taps 1: the row itself (-1: past M); taps 9: first pixel of the row's image
#pragma unroll
    for (int e = 0; e < EA; ++e) {
        const int m = m0 + lrow + 64 * e;
        if (p.taps == 9) {
            const int nb = m / hw, r = m - nb * hw, oy = r / p.Wo, ox = r - oy * p.Wo;
            iy0[e] = m < p.M ? oy * p.stride + p.coff - 1 : -(1 << 20);
            ix0[e] = ox * p.stride + p.coff - 1;
            base[e] = (long long)nb * p.Hs * p.Ws;
        } else {
            iy0[e] = ix0[e] = 0;
            base[e] = m < p.M ? m : -1;
        }
    }
    floatx4 ra[EA], rb[EB];
    auto fetch = [&](int k0) {
        int tap = 0, c = k0;
        if (p.taps == 9) {
            const int chunk = k0 / (9 * ck), rem = k0 - chunk * 9 * ck;
            tap = rem / ck;
            c = chunk * ck + (rem - tap * ck);
        }
        const bool first = c < p.C0;
        const float* src = first ? A0 + c + kq : A1 + (c - p.C0) + kq;
        const int ld = first ? p.lda0 : p.lda1;
        const int dy = tap / 3, dx = tap - 3 * dy;
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            long long pix = base[e];
            bool ok = pix >= 0;
            if (p.taps == 9) {
                const int iy = iy0[e] + dy, ix = ix0[e] + dx;
                ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                pix += (long long)(iy >> p.ups) * p.Ws + (ix >> p.ups);
            }
            ra[e] = ok ? *(const floatx4*)(src + pix * ld) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int n = n0 + lrow + 64 * e;
            rb[e] = n < p.Npad ? *(const floatx4*)(Wp + (long long)n * p.K + k0 + kq) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int e = 0; e < EA; ++e) *(floatx4*)&As[buf][(lrow + 64 * e) * LDT + kq] = ra[e];
#pragma unroll
        for (int e = 0; e < EB; ++e) *(floatx4*)&Bs[buf][(lrow + 64 * e) * LDT + kq] = rb[e];
    };

    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int wm = (wv >> 1) * (BMT / 2), wn = (wv & 1) * (BNT / 2);
    floatx4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash(0);
    __syncthreads();
    const int nk = p.K / 16;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * 16);
        floatx4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const floatx4*)&As[buf][(wm + i * 16 + li) * LDT + lg * 4];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const floatx4*)&Bs[buf][(wn + j * 16 + li) * LDT + lg * 4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        if (kt + 1 < nk) stash(buf ^ 1);
        __syncthreads();
    }
    // accumulator layout of the 16x16 MFMA: lane (li, lg), register r = element (row 4 lg + r, column li)
    const long long zD = z * p.sD, zR = z * p.sR;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn + j * 16 + li;
            if (n >= p.N) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm + i * 16 + lg * 4 + r;
                if (m >= p.M) continue;
                float x = acc[i][j][r] * p.alpha + bv;
                if (p.epi == EPI_STD) {
                    if (p.R) x += p.R[zR + (long long)m * p.ldr + n];
                    p.D[zD + (long long)m * p.ldd + n] = x;
                } else if (p.epi == EPI_NCHW_F32) {
                    const int nb = m / hw, pix = m - nb * hw;
                    p.Df[zD + ((long long)nb * p.N + n) * hw + pix] = x;
                } else if (p.epi == EPI_IMAGE) {
                    x = x / 2.0f + 0.5f;
                    p.Df[zD + (long long)m * p.N + n] = fminf(fmaxf(x, 0.0f), 1.0f);
                } else {   // EPI_TRANS
                    p.D[zD + (long long)n * p.ldd + m] = x;
                }
            }
        }
}

// GEGLU on an un-fused projection: Y[m][j] = X[m][j] * gelu(X[m][half + j]) (first half = value, second half = gate)
__global__ void geglu_f32_kernel(const float* X, float* Y, long long rows, int half_n, int tanh_form)
{
    const long long total = rows * half_n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long m = i / half_n;
        const int j = (int)(i - m * half_n);
        const float a = X[m * 2 * half_n + j], g = X[m * 2 * half_n + half_n + j];
        const float ge = tanh_form ? gelu_tanh_form(g) : 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));
        Y[i] = a * ge;
    }
}

// ------------------------------------------------------------------------------------------ attention (two passes per query)
struct AttnF { const float* Q; const float* K; const float* V; float* O; int ldq, ldk, ldv, ldo, NB, heads, Sq, Skv, d; float scale; };
// one wave per query: lanes split the keys; scores are recomputed in the second pass instead of stored
__global__ __launch_bounds__(256) void attn_f32_kernel(AttnF p)
{
    const int lane = threadIdx.x & 63;
    const long long gq = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gq >= (long long)p.NB * p.heads * p.Sq) return;
    const int q = (int)(gq % p.Sq), head = (int)((gq / p.Sq) % p.heads), nb = (int)(gq / ((long long)p.Sq * p.heads));
    const int d = p.d;
    const float* Qr = p.Q + ((long long)nb * p.Sq + q) * p.ldq + head * d;
    const float* Kb = p.K + (long long)nb * p.Skv * p.ldk + head * d;
    const float* Vb = p.V + (long long)nb * p.Skv * p.ldv + head * d;
    float* Or = p.O + ((long long)nb * p.Sq + q) * p.ldo + head * d;
    float mx = -INFINITY;
    for (int k = lane; k < p.Skv; k += 64) {
        float s = 0.f;
        for (int c = 0; c < d; ++c) s = fmaf(Qr[c], Kb[(long long)k * p.ldk + c], s);
        mx = fmaxf(mx, s * p.scale);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    // each lane owns channels lane, lane + 64, lane + 128 (d <= 192) and walks all keys; the softmax weights are shared through shuffles
    float acc[3] = {0.f, 0.f, 0.f}, l = 0.f;
    for (int k0 = 0; k0 < p.Skv; k0 += 64) {
        const int k = k0 + lane;
        float e = 0.f;
        if (k < p.Skv) {
            float s = 0.f;
            for (int c = 0; c < d; ++c) s = fmaf(Qr[c], Kb[(long long)k * p.ldk + c], s);
            e = expf(s * p.scale - mx);
        }
        l += e;
        const int kn = min(64, p.Skv - k0);
        for (int j = 0; j < kn; ++j) {
            const float ej = __shfl(e, j);
            const float* vr = Vb + (long long)(k0 + j) * p.ldv;
#pragma unroll
            for (int u = 0; u < 3; ++u) { const int c = lane + u * 64; if (c < d) acc[u] = fmaf(ej, vr[c], acc[u]); }
        }
    }
    l = wsum(l);
    const float inv = 1.0f / l;
#pragma unroll
    for (int u = 0; u < 3; ++u) { const int c = lane + u * 64; if (c < d) Or[c] = acc[u] * inv; }
}
// wide heads (the VAE's single 512-channel head): same scheme, channels in a loop
__global__ __launch_bounds__(256) void attn_f32_wide_kernel(AttnF p)
{
    extern __shared__ float sacc[];   // [4 waves][d]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long gq = (long long)blockIdx.x * 4 + wv;
    if (gq >= (long long)p.NB * p.heads * p.Sq) return;
    const int q = (int)(gq % p.Sq), head = (int)((gq / p.Sq) % p.heads), nb = (int)(gq / ((long long)p.Sq * p.heads));
    const int d = p.d;
    const float* Qr = p.Q + ((long long)nb * p.Sq + q) * p.ldq + head * d;
    const float* Kb = p.K + (long long)nb * p.Skv * p.ldk + head * d;
    const float* Vb = p.V + (long long)nb * p.Skv * p.ldv + head * d;
    float* Or = p.O + ((long long)nb * p.Sq + q) * p.ldo + head * d;
    float* acc = sacc + (size_t)wv * d;
    for (int c = lane; c < d; c += 64) acc[c] = 0.f;
    float mx = -INFINITY;
    for (int k = lane; k < p.Skv; k += 64) {
        float s = 0.f;
        for (int c = 0; c < d; ++c) s = fmaf(Qr[c], Kb[(long long)k * p.ldk + c], s);
        mx = fmaxf(mx, s * p.scale);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float l = 0.f;
    for (int k0 = 0; k0 < p.Skv; k0 += 64) {
        const int k = k0 + lane;
        float e = 0.f;
        if (k < p.Skv) {
            float s = 0.f;
            for (int c = 0; c < d; ++c) s = fmaf(Qr[c], Kb[(long long)k * p.ldk + c], s);
            e = expf(s * p.scale - mx);
        }
        l += e;
        const int kn = min(64, p.Skv - k0);
        for (int j = 0; j < kn; ++j) {
            const float ej = __shfl(e, j);
            const float* vr = Vb + (long long)(k0 + j) * p.ldv;
            for (int c = lane; c < d; c += 64) acc[c] = fmaf(ej, vr[c], acc[c]);   // each lane touches only its own channels
        }
    }
    l = wsum(l);
    const float inv = 1.0f / l;
    for (int c = lane; c < d; c += 64) Or[c] = acc[c] * inv;
}

// ------------------------------------------------------------------------------------------ GroupNorm: one block per (image, group)
__global__ __launch_bounds__(256) void gn_f32_kernel(const float* X0, const float* X1, int C0, int C1, int HW, int groups, float eps,
                                                     const float* gamma, const float* beta, int silu, float* Y)
{
    __shared__ float red[4];
    const int C = C0 + C1, cpg = C / groups, g = blockIdx.x, nb = blockIdx.y, tid = threadIdx.x;
    const int total = HW * cpg;
    auto at = [&](int i) -> float {
        const int px = i / cpg, c = g * cpg + (i - px * cpg);
        const long long row = (long long)nb * HW + px;
        return c < C0 ? X0[row * C0 + c] : X1[row * C1 + (c - C0)];
    };
    auto block_sum = [&](float v) -> float {
        v = wsum(v);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float s = 0.f;
    for (int i = tid; i < total; i += 256) s += at(i);
    const float mean = block_sum(s) / (float)total;
    float q = 0.f;
    for (int i = tid; i < total; i += 256) { const float dlt = at(i) - mean; q += dlt * dlt; }
    const float rstd = rsqrtf(block_sum(q) / (float)total + eps);
    for (int i = tid; i < total; i += 256) {
        const int px = i / cpg, c = g * cpg + (i - px * cpg);
        float y = (at(i) - mean) * rstd * gamma[c] + beta[c];
        if (silu) y = y / (1.0f + expf(-y));
        Y[((long long)nb * HW + px) * C + c] = y;
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm (+ fp32 sum), one wave per row
__global__ __launch_bounds__(256) void ln_f32_kernel(const float* S, const float* A, const float* B, float* Sout, float* Y, int rows, int C,
                                                     const float* gamma, const float* beta, float eps)
{
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const long long off = row * C;
    auto at = [&](int c) -> float { return (S ? S[off + c] : 0.f) + (A ? A[off + c] : 0.f) + (B ? B[off + c] : 0.f); };
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += at(c);
    const float mean = wsum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float dlt = at(c) - mean; q += dlt * dlt; }
    const float rstd = rsqrtf(wsum(q) / (float)C + eps);
    for (int c = lane; c < C; c += 64) {
        const float x = at(c);
        if (Sout) Sout[off + c] = x;
        Y[off + c] = (x - mean) * rstd * gamma[c] + beta[c];
    }
}
__global__ void sum3_f32_kernel(const float* S, const float* A, const float* B, float* Y, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) Y[i] = S[i] + A[i] + B[i];
}

// ------------------------------------------------------------------------------------------ packing
__global__ void pack_nchw_f32_kernel(const float* x, float* y, int NB, int C, int HW, int Cpad, float scale)
{
    const long long total = (long long)NB * HW * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long long row = i / Cpad;
        const int nb = (int)(row / HW), pix = (int)(row - (long long)nb * HW);
        y[i] = c < C ? x[((long long)nb * C + c) * HW + pix] * scale : 0.f;
    }
}
__global__ void pack_latents_f32_kernel(const float* lat, float* packed, int B, int dup, int C, int HW, int Cpad, float scale)
{
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW), c = (int)((i / HW) % C), b = (int)(i / ((long long)HW * C));
        const float v = lat[i] * scale;
        for (int r = 0; r < dup; ++r) packed[(((long long)(b + r * B)) * HW + pix) * Cpad + c] = v;
    }
}
__global__ void gemv_f32_kernel(const float* W, const float* x, const float* bias, float* y, int N, int K, int pre_silu, int post_silu)
{
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = x[k];
        if (pre_silu) xv = xv / (1.0f + expf(-xv));
        acc = fmaf(W[(long long)n * K + k], xv, acc);
    }
    acc = wsum(acc);
    if (lane == 0) {
        float v = acc + (bias ? bias[n] : 0.f);
        if (post_silu) v = v / (1.0f + expf(-v));
        y[n] = v;
    }
}
}  // namespace

// ------------------------------------------------------------------------------------------ launchers (pointers declared half_t* address floats)
void f32_launch_gemm(const LaunchCtx& ctx, const GemmArgs& a)
{
    SDE_REQUIRE(a.epi == EPI_STD || a.epi == EPI_NCHW_F32 || a.epi == EPI_IMAGE || a.epi == EPI_TRANS, SDE_ERR_INVALID_ARGUMENT,
                "fp32 engine: fused GEGLU epilogues are not used (the projection is followed by f32_launch_geglu)");
    SDE_REQUIRE(!a.Rlo && !a.Dlo, SDE_ERR_INVALID_ARGUMENT, "fp32 engine: no split-fp16 tensors");
    GemmF p;
    p.A0 = (const float*)a.A0; p.A1 = (const float*)a.A1; p.C0 = a.C0; p.C1 = a.C1; p.lda0 = a.lda0; p.lda1 = a.lda1;
    p.M = a.M; p.N = a.N; p.K = a.K; p.taps = a.taps; p.stride = a.stride; p.coff = a.coff; p.ups = a.ups; p.Hi = a.Hi; p.Wi = a.Wi; p.Ho = a.Ho; p.Wo = a.Wo;
    p.Hs = a.ups ? (a.Hi + 1) / 2 : a.Hi; p.Ws = a.ups ? (a.Wi + 1) / 2 : a.Wi;
    p.W = (const float*)a.W; p.Npad = a.Npad; p.bias = a.bias; p.R = (const float*)a.R; p.ldr = a.ldr;
    p.D = (float*)a.D; p.Df = a.Df; p.ldd = a.ldd; p.epi = a.epi; p.alpha = a.alpha; p.sA = a.sA; p.sW = a.sW; p.sD = a.sD; p.sR = a.sR;
    ProfScope ps(ctx, a.taps == 9 ? "conv3x3_f32" : "gemm_f32", 2.0 * a.M * (double)a.N * a.K * a.batch,
                 4.0 * a.batch * ((double)a.M * a.K / (a.taps == 9 ? 9.0 : 1.0) + (double)a.N * a.K + (double)a.M * a.N));
    // MFMA tiles need 16-float slabs that stay inside one tap / chunk / concat source, and 16-byte aligned rows
    const int Ctot = a.C0 + a.C1, ckk = (Ctot % 64 == 0) ? 64 : Ctot;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    static const int mfma_env = [] { const char* v = getenv("SDE_F32_MFMA"); return v ? atoi(v) : 1; }();   // 0: plain FMA kernel everywhere
    const bool mfma = mfma_env && !ctx.force_naive && a.K % 16 == 0 && a.C0 % 16 == 0 && a.C1 % 16 == 0 && (a.taps == 1 || ckk % 16 == 0) &&
                      a.lda0 % 4 == 0 && (!a.A1 || a.lda1 % 4 == 0) && al16(a.A0) && al16(a.A1) && al16(a.W) && a.sA % 4 == 0 && a.sW % 4 == 0;
    if (!mfma) hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(a.M, BM), cdiv(a.N, BN), a.batch), dim3(256), 0, ctx.stream, p);
    else if ((long long)cdiv(a.M, 128) * cdiv(a.N, 128) * a.batch >= 192)
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<128, 128>), dim3(cdiv(a.M, 128), cdiv(a.N, 128), a.batch), dim3(256), 0, ctx.stream, p);
    else
        hipLaunchKernelGGL((gemm_f32_mfma_kernel<64, 64>), dim3(cdiv(a.M, 64), cdiv(a.N, 64), a.batch), dim3(256), 0, ctx.stream, p);
    HIP_CHECK(hipGetLastError());
}

void f32_launch_geglu(const LaunchCtx& ctx, const half_t* X, half_t* Y, int64_t rows, int half_n, int tanh_form)
{
    ProfScope ps(ctx, "geglu_f32", 0, 12.0 * rows * half_n);
    hipLaunchKernelGGL(geglu_f32_kernel, dim3((unsigned)std::min<long long>(cdiv64(rows * half_n, 256), 8192)), dim3(256), 0, ctx.stream,
                       (const float*)X, (float*)Y, (long long)rows, half_n, tanh_form);
    HIP_CHECK(hipGetLastError());
}

void f32_launch_attention(const LaunchCtx& ctx, const AttnArgs& a)
{
    AttnF p{(const float*)a.Q, (const float*)a.K, (const float*)a.V, (float*)a.O, a.ldq, a.ldk, a.ldv, a.ldo, a.NB, a.heads, a.Sq, a.Skv, a.d, a.scale};
    ProfScope ps(ctx, "attn_f32", 4.0 * a.NB * a.heads * (double)a.Sq * a.Skv * a.d, 0);
    const long long total = (long long)a.NB * a.heads * a.Sq;
    if (a.d <= 192) hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)cdiv64(total, 4)), dim3(256), 0, ctx.stream, p);
    else {
        SDE_REQUIRE((size_t)a.d * 16 <= 64 * 1024, SDE_ERR_INVALID_ARGUMENT, "fp32 attention: head too wide");
        hipLaunchKernelGGL(attn_f32_wide_kernel, dim3((unsigned)cdiv64(total, 4)), dim3(256), (size_t)a.d * 16, ctx.stream, p);
    }
    HIP_CHECK(hipGetLastError());
}

void f32_launch_groupnorm(const LaunchCtx& ctx, const GroupNormArgs& a)
{
    const int C = a.C0 + a.C1;
    SDE_REQUIRE(C % a.groups == 0, SDE_ERR_INVALID_ARGUMENT, "groupnorm: C must be a multiple of the group count");
    ProfScope ps(ctx, "groupnorm_f32", 0, 16.0 * a.NB * a.HW * C);
    hipLaunchKernelGGL(gn_f32_kernel, dim3(a.groups, a.NB), dim3(256), 0, ctx.stream, (const float*)a.X0, (const float*)a.X1, a.C0, a.C1, a.HW,
                       a.groups, a.eps, a.gamma, a.beta, a.silu ? 1 : 0, (float*)a.Y);
    HIP_CHECK(hipGetLastError());
}

void f32_launch_layernorm_sum(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, float* S32_out, half_t* Y, int rows, int C,
                              const float* gamma, const float* beta, float eps)
{
    ProfScope ps(ctx, "layernorm_f32", 0, 8.0 * rows * C);
    hipLaunchKernelGGL(ln_f32_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, ctx.stream, S32, (const float*)A, (const float*)B, S32_out, (float*)Y, rows, C,
                       gamma, beta, eps);
    HIP_CHECK(hipGetLastError());
}

void f32_launch_sum_round(const LaunchCtx& ctx, const float* S32, const half_t* A, const half_t* B, half_t* Y, int64_t n)
{
    hipLaunchKernelGGL(sum3_f32_kernel, dim3((unsigned)std::min<long long>(cdiv64(n, 256), 8192)), dim3(256), 0, ctx.stream, S32, (const float*)A,
                       (const float*)B, (float*)Y, (long long)n);
    HIP_CHECK(hipGetLastError());
}

void f32_launch_pack_nchw(const LaunchCtx& ctx, const float* x, half_t* y, int NB, int C, int H, int W, int Cpad, float scale)
{
    const long long total = (long long)NB * H * W * Cpad;
    hipLaunchKernelGGL(pack_nchw_f32_kernel, dim3((unsigned)std::min<long long>(cdiv64(total, 256), 4096)), dim3(256), 0, ctx.stream, x, (float*)y, NB, C,
                       H * W, Cpad, scale);
    HIP_CHECK(hipGetLastError());
}

void f32_launch_pack_latents(const LaunchCtx& ctx, const float* latents, half_t* packed, int B, int dup, int C, int H, int W, int Cpad, float scale)
{
    const long long total = (long long)B * C * H * W;
    hipLaunchKernelGGL(pack_latents_f32_kernel, dim3((unsigned)std::min<long long>(cdiv64(total, 256), 2048)), dim3(256), 0, ctx.stream, latents,
                       (float*)packed, B, dup, C, H * W, Cpad, scale);
    HIP_CHECK(hipGetLastError());
}

void f32_launch_gemv(const LaunchCtx& ctx, const half_t* W, const float* x, const float* bias, float* y, int N, int K, bool pre_silu, bool post_silu)
{
    hipLaunchKernelGGL(gemv_f32_kernel, dim3(cdiv(N, 4)), dim3(256), 0, ctx.stream, (const float*)W, x, bias, y, N, K, pre_silu ? 1 : 0, post_silu ? 1 : 0);
    HIP_CHECK(hipGetLastError());
}
