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
    int splitk; float* ws;     // MFMA kernel, batch 1: splitk > 1 = blockIdx.z owns a K range and leaves a raw fp32 partial tile in ws[z][M][N]
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
    const int z = p.splitk > 1 ? 0 : blockIdx.z;
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

    int kt0 = 0, nk = p.K / 16;
    if (p.splitk > 1) {
        const int per = (nk + p.splitk - 1) / p.splitk;
        kt0 = blockIdx.z * per;
        nk = min(nk, kt0 + per);                       // the launcher keeps every range non-empty
    }
    fetch(kt0 * 16);
    stash(kt0 & 1);
    __syncthreads();
    for (int kt = kt0; kt < nk; ++kt) {
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
    if (p.splitk > 1) {
        float* wsz = p.ws + (long long)blockIdx.z * p.M * p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn + j * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm + i * 16 + lg * 4 + r;
                    if (m < p.M && n < p.N) wsz[(long long)m * p.N + n] = acc[i][j][r];
                }
            }
        return;
    }
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

// split-K tail: adds the partial tiles in slice order (a fixed order: bit-reproducible) and applies the epilogue
__global__ __launch_bounds__(256) void gemm_f32_splitk_reduce_kernel(GemmF p)
{
    const long long total = (long long)p.M * p.N;
    const int hw = p.Ho * p.Wo;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / p.N), n = (int)(i - (long long)m * p.N);
        float a = 0.f;
        for (int z = 0; z < p.splitk; ++z) a += p.ws[z * total + i];
        float x = a * p.alpha + (p.bias ? p.bias[n] : 0.f);
        if (p.epi == EPI_STD) {
            if (p.R) x += p.R[(long long)m * p.ldr + n];
            p.D[(long long)m * p.ldd + n] = x;
        } else if (p.epi == EPI_NCHW_F32) {
            const int nb = m / hw, pix = m - nb * hw;
            p.Df[((long long)nb * p.N + n) * hw + pix] = x;
        } else if (p.epi == EPI_IMAGE) {
            x = x / 2.0f + 0.5f;
            p.Df[(long long)m * p.N + n] = fminf(fmaxf(x, 0.0f), 1.0f);
        } else {
            p.D[(long long)n * p.ldd + m] = x;
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

// MFMA attention in fp32 (v_mfma_f32_16x16x4_f32), online softmax.  One block = 4 waves x 32 queries of one (image, head); keys in tiles
// of 32 staged through LDS as [key][DT*16 + 4] floats for both K and V (columns d .. DT*16 are zero), the next tile's global loads in
// flight over the current tile's MFMAs.
//   S^T[key][q] = K . Q^T : A = K rows (one b128 LDS read per 16 keys and 16-wide chunk of d: the lane's four values feed four MFMAs, the
//                 same k-permutation as gemm_f32_mfma_kernel), B = the wave's Q fragments (registers, pre-multiplied by scale * log2 e).
//                 The result leaves lane (li = query, lg) holding keys 4 lg + r: exactly the B-operand layout of the second product
//   O^T[c][q]   = V^T . P^T: A = V[key = 4 lg + s][c = li] (four b32 LDS reads, row stride = 16 banks mod 64: conflict-free), B = P.
// Row maxima are shared over the four lg groups with two shuffles; row sums are kept per lane and reduced once at the end.
template <int DT>
__global__ __launch_bounds__(256) void attn_f32_mfma_kernel(AttnF p)
{
    constexpr int KT = 32, LDK = DT * 16 + 4, QT = 2;
    constexpr int NLD = (KT * DT * 4 + 255) / 256;            // float4 loads per thread, tile and operand
    __shared__ __align__(16) float Ks[KT * LDK];
    __shared__ __align__(16) float Vs[KT * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, lg = lane >> 4;
    const int qblocks = (p.Sq + 127) / 128;
    const int qb = blockIdx.x % qblocks, head = (blockIdx.x / qblocks) % p.heads, nb = blockIdx.x / (qblocks * p.heads);
    const int d = p.d, d4 = d >> 2;
    const float* Kb = p.K + (long long)nb * p.Skv * p.ldk + head * d;
    const float* Vb = p.V + (long long)nb * p.Skv * p.ldv + head * d;
    for (int i = tid; i < KT * LDK; i += 256) { Ks[i] = 0.f; Vs[i] = 0.f; }

    const float qscale = p.scale * 1.44269504088896340736f;
    const int q0 = qb * 128 + wv * 32;
    floatx4 qf[QT][DT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int q = q0 + t * 16 + li;
        const float* Qr = p.Q + ((long long)nb * p.Sq + min(q, p.Sq - 1)) * p.ldq + head * d;
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            const int col = c * 16 + lg * 4;
            floatx4 v = col < d ? *(const floatx4*)(Qr + col) : floatx4{0.f, 0.f, 0.f, 0.f};
            qf[t][c] = v * qscale;
        }
    }
    floatx4 o[QT][DT];
    float mx[QT], lsum[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        mx[t] = -INFINITY; lsum[t] = 0.f;
#pragma unroll
        for (int c = 0; c < DT; ++c) o[t][c] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    floatx4 rk[NLD], rv[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int id = tid + e * 256, key = id / d4, c4 = id - key * d4;
            const bool ok = key < KT && k0 + key < p.Skv;
            rk[e] = ok ? *(const floatx4*)(Kb + (long long)(k0 + key) * p.ldk + c4 * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
            rv[e] = ok ? *(const floatx4*)(Vb + (long long)(k0 + key) * p.ldv + c4 * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int id = tid + e * 256, key = id / d4, c4 = id - key * d4;
            if (key < KT) { *(floatx4*)&Ks[key * LDK + c4 * 4] = rk[e]; *(floatx4*)&Vs[key * LDK + c4 * 4] = rv[e]; }
        }
    };
    fetch(0);
    __syncthreads();                       // zero fill done
    for (int k0 = 0; k0 < p.Skv; k0 += KT) {
        stash();
        __syncthreads();
        if (k0 + KT < p.Skv) fetch(k0 + KT);
        floatx4 s[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) { s[t][0] = floatx4{0.f, 0.f, 0.f, 0.f}; s[t][1] = floatx4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                const floatx4 kf = *(const floatx4*)&Ks[(ks * 16 + li) * LDK + c * 16 + lg * 4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < QT; ++t) s[t][ks] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[u], qf[t][c][u], s[t][ks], 0, 0, 0);
            }
        // lane (li, lg) holds, for query li of each q-tile, keys k0 + ks*16 + 4 lg + r
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float tm = -INFINITY;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (k0 + ks * 16 + lg * 4 + r >= p.Skv) s[t][ks][r] = -INFINITY;
                    tm = fmaxf(tm, s[t][ks][r]);
                }
            tm = fmaxf(tm, __shfl_xor(tm, 16));
            tm = fmaxf(tm, __shfl_xor(tm, 32));
            const float mnew = fmaxf(mx[t], tm);               // finite: every tile holds at least one valid key
            const float alpha = exp2f(mx[t] - mnew);
            mx[t] = mnew;
            float ps = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[t][ks][r] = exp2f(s[t][ks][r] - mnew); ps += s[t][ks][r]; }
            lsum[t] = lsum[t] * alpha + ps;
#pragma unroll
            for (int c = 0; c < DT; ++c) o[t][c] *= alpha;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int c = 0; c < DT; ++c) {
                float vf[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) vf[u] = Vs[(ks * 16 + lg * 4 + u) * LDK + c * 16 + li];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < QT; ++t) o[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[u], s[t][ks][u], o[t][c], 0, 0, 0);
            }
        __syncthreads();
    }
    // O^T accumulator: lane (li = query, lg), register r = channel c*16 + 4 lg + r
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l = lsum[t];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const float inv = 1.0f / l;
        const int q = q0 + t * 16 + li;
        if (q >= p.Sq) continue;
        float* Or = p.O + ((long long)nb * p.Sq + q) * p.ldo + head * d;
#pragma unroll
        for (int c = 0; c < DT; ++c) {
            const int col = c * 16 + lg * 4;
            if (col < d) *(floatx4*)(Or + col) = o[t][c] * inv;
        }
    }
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

// Row-streaming GroupNorm (the plain kernel above puts one block on each (image, group): 64 blocks at UNet batch 2).  Thread = one
// float4 of channels, fixed for the thread's life, walking rows: coalesced row reads, per-channel sums in fp64 registers (single pass:
// sum and sum of squares), combined per group through LDS in a fixed order.  Stage 1 leaves [image][slice][group] partials, stage 2
// adds a group's slices in slice order, so the result does not depend on scheduling.
struct GnF { const float* X0; const float* X1; int C0, C1, HW, groups, slices, rows_per_slice, rpi; float eps; const float* gamma; const float* beta;
             int silu; float* Y; double* part; };
__global__ void gn_f32_rows_stats_kernel(GnF p)
{
    extern __shared__ double sh[];                      // [rpi][C][2]
    const int C = p.C0 + p.C1, C4 = C >> 2, cpg = C / p.groups;
    const int tid = threadIdx.x, c4 = tid % C4, r0 = tid / C4, c = c4 * 4;
    const int nb = blockIdx.y, sl = blockIdx.x;
    const int row_end = min(p.HW, (sl + 1) * p.rows_per_slice);
    const bool first = c < p.C0;
    const float* src = first ? p.X0 + (long long)nb * p.HW * p.C0 + c : p.X1 + (long long)nb * p.HW * p.C1 + (c - p.C0);
    const int ld = first ? p.C0 : p.C1;
    double sm[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    for (int r = sl * p.rows_per_slice + r0; r < row_end; r += p.rpi) {
        const floatx4 v = *(const floatx4*)(src + (long long)r * ld);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double x = v[j]; sm[j] += x; sq[j] = fma(x, x, sq[j]); }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { sh[((size_t)r0 * C + c + j) * 2] = sm[j]; sh[((size_t)r0 * C + c + j) * 2 + 1] = sq[j]; }
    __syncthreads();
    if (tid < p.groups) {
        double a = 0, b = 0;
        for (int r = 0; r < p.rpi; ++r)
            for (int k = 0; k < cpg; ++k) { a += sh[((size_t)r * C + tid * cpg + k) * 2]; b += sh[((size_t)r * C + tid * cpg + k) * 2 + 1]; }
        double* o = p.part + (((size_t)nb * p.slices + sl) * p.groups + tid) * 2;
        o[0] = a; o[1] = b;
    }
}
__global__ void gn_f32_rows_apply_kernel(GnF p)
{
    extern __shared__ double sh[];                      // reused as float [groups][2]: mean, rstd
    float* st = (float*)sh;
    const int C = p.C0 + p.C1, C4 = C >> 2, cpg = C / p.groups;
    const int tid = threadIdx.x, c4 = tid % C4, r0 = tid / C4, c = c4 * 4;
    const int nb = blockIdx.y, sl = blockIdx.x;
    if (tid < p.groups) {
        double a = 0, b = 0;
        for (int k = 0; k < p.slices; ++k) {
            const double* o = p.part + (((size_t)nb * p.slices + k) * p.groups + tid) * 2;
            a += o[0]; b += o[1];
        }
        const double n = (double)p.HW * cpg, mean = a / n, var = fmax(b / n - mean * mean, 0.0);
        st[tid * 2] = (float)mean;
        st[tid * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    const bool first = c < p.C0;
    const float* src = first ? p.X0 + (long long)nb * p.HW * p.C0 + c : p.X1 + (long long)nb * p.HW * p.C1 + (c - p.C0);
    const int ld = first ? p.C0 : p.C1;
    float mean[4], rs[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int g = (c + j) / cpg; mean[j] = st[g * 2]; rs[j] = st[g * 2 + 1]; ga[j] = p.gamma[c + j]; be[j] = p.beta[c + j]; }
    const int row_end = min(p.HW, (sl + 1) * p.rows_per_slice);
    for (int r = sl * p.rows_per_slice + r0; r < row_end; r += p.rpi) {
        const floatx4 v = *(const floatx4*)(src + (long long)r * ld);
        floatx4 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = (v[j] - mean[j]) * rs[j] * ga[j] + be[j];
            if (p.silu) t = t / (1.0f + expf(-t));
            y[j] = t;
        }
        *(floatx4*)(p.Y + ((long long)nb * p.HW + r) * C + c) = y;
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
    p.splitk = 1; p.ws = nullptr;
    if (!mfma) { hipLaunchKernelGGL(gemm_f32_kernel, dim3(cdiv(a.M, BM), cdiv(a.N, BN), a.batch), dim3(256), 0, ctx.stream, p); HIP_CHECK(hipGetLastError()); return; }
    // tuning hooks: SDE_F32_TILE = 64 / 128 forces the block tile, SDE_F32_SPLITK = n the K split (where it applies)
    static const int tile_env = [] { const char* v = getenv("SDE_F32_TILE"); return v ? atoi(v) : 0; }();
    static const int sk_env = [] { const char* v = getenv("SDE_F32_SPLITK"); return v ? atoi(v) : 0; }();
    // Few tiles, long K (UNet batch 2: the 8 x 8 level's 3x3 convs are 40 tiles of 720 slabs): split K so that ~3 blocks per CU exist;
    // partial tiles through the split-K workspace, reduced in slice order.  The 128 x 128 tile (half the operand traffic per flop) is
    // taken when, with its split, it still yields two blocks per CU; measured per shape in DESIGN.md section 4 (fp32 engine).
    const int nk = a.K / 16;
    const bool can_split = a.batch == 1 && ctx.splitk_ws && nk >= 32;
    auto split_for = [&](long long tiles) {
        int sk = 1;
        if (can_split && tiles < 512) {
            sk = sk_env ? sk_env : (int)std::min<long long>(cdiv(768, (int)tiles), 16);
            sk = std::max(1, std::min(sk, nk / 16));
            while (sk > 1 && (size_t)sk * a.M * a.N * sizeof(float) > ctx.splitk_ws_bytes) --sk;
            if (sk > 1) sk = cdiv(nk, cdiv(nk, sk));               // no empty K range
        }
        return sk;
    };
    const long long t128 = (long long)cdiv(a.M, 128) * cdiv(a.N, 128) * a.batch, t64 = (long long)cdiv(a.M, 64) * cdiv(a.N, 64) * a.batch;
    const int tile = tile_env ? tile_env : (t128 * split_for(t128) >= 512 ? 128 : 64);
    const int sk = split_for(tile == 128 ? t128 : t64);
    if (sk > 1) { p.splitk = sk; p.ws = ctx.splitk_ws; }
    const dim3 grid(cdiv(a.M, tile), cdiv(a.N, tile), sk > 1 ? sk : a.batch);
    if (tile == 128) hipLaunchKernelGGL((gemm_f32_mfma_kernel<128, 128>), grid, dim3(256), 0, ctx.stream, p);
    else hipLaunchKernelGGL((gemm_f32_mfma_kernel<64, 64>), grid, dim3(256), 0, ctx.stream, p);
    if (sk > 1)
        hipLaunchKernelGGL(gemm_f32_splitk_reduce_kernel, dim3((unsigned)std::min<long long>(cdiv64((long long)a.M * a.N, 256), 2048)), dim3(256), 0,
                           ctx.stream, p);
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
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    static const int mfma_env = [] { const char* v = getenv("SDE_F32_MFMA"); return v ? atoi(v) : 1; }();
    const bool mfma = mfma_env && !ctx.force_naive && a.d % 4 == 0 && a.d <= 160 && a.Skv >= 1 && a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 &&
                      a.ldo % 4 == 0 && al16(a.Q) && al16(a.K) && al16(a.V) && al16(a.O);
    if (mfma) {
        const dim3 grid((unsigned)((long long)a.NB * a.heads * cdiv(a.Sq, 128)));
        if (a.d <= 48) hipLaunchKernelGGL(attn_f32_mfma_kernel<3>, grid, dim3(256), 0, ctx.stream, p);
        else if (a.d <= 80) hipLaunchKernelGGL(attn_f32_mfma_kernel<5>, grid, dim3(256), 0, ctx.stream, p);
        else hipLaunchKernelGGL(attn_f32_mfma_kernel<10>, grid, dim3(256), 0, ctx.stream, p);
    } else if (a.d <= 192) hipLaunchKernelGGL(attn_f32_kernel, dim3((unsigned)cdiv64(total, 4)), dim3(256), 0, ctx.stream, p);
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
    const int C4 = C / 4;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if (!ctx.force_naive && C % 4 == 0 && a.C0 % 4 == 0 && C4 <= 1024 && a.groups <= 64 && a.ws && al16(a.X0) && al16(a.X1) && al16(a.Y) &&
        ((uintptr_t)a.ws & 7) == 0) {
        GnF g;
        g.rpi = C4 >= 256 ? 1 : 256 / C4;
        const int threads = g.rpi * C4;
        g.slices = std::max(1, std::min(64, a.HW / (g.rpi * 4)));          // >= 4 rows per thread, <= 64 slices (groupnorm_ws_floats)
        g.rows_per_slice = cdiv(a.HW, g.slices);
        g.slices = cdiv(a.HW, g.rows_per_slice);
        g.X0 = (const float*)a.X0; g.X1 = (const float*)a.X1; g.C0 = a.C0; g.C1 = a.C1; g.HW = a.HW; g.groups = a.groups; g.eps = a.eps;
        g.gamma = a.gamma; g.beta = a.beta; g.silu = a.silu ? 1 : 0; g.Y = (float*)a.Y; g.part = (double*)a.ws;
        const size_t lds = (size_t)g.rpi * C * 2 * sizeof(double);         // <= 16 * C bytes * rpi: 64 KB at C = 4096
        allow_dynamic_lds((const void*)gn_f32_rows_stats_kernel, 64 * 1024);
        hipLaunchKernelGGL(gn_f32_rows_stats_kernel, dim3(g.slices, a.NB), dim3(threads), lds, ctx.stream, g);
        hipLaunchKernelGGL(gn_f32_rows_apply_kernel, dim3(g.slices, a.NB), dim3(threads), 64 * 2 * sizeof(double), ctx.stream, g);
        HIP_CHECK(hipGetLastError());
        return;
    }
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
