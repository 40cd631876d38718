// gemm_common.h — device-side argument block and the shared epilogue of the GEMM / implicit-GEMM conv kernels
// (gemm.hip: 4-wave 128/64-row tiles; gemm_big.hip: 8-wave 256x320 ping-pong tiles).
#pragma once
#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

struct GemmDev {
    const half_t* A0; const half_t* A1;
    int C0, C1, lda0, lda1;
    int M, N, K, taps, stride, ups, Hi, Wi, Ho, Wo, Hs, Ws, coff;
    const half_t* W; int Npad;
    const float* bias; const half_t* R; const half_t* Rlo; half_t* Dlo; int ldr;
    half_t* D; float* Df; int ldd; int epi; int gelu_tanh; float alpha;
    long long sA, sW, sD, sR;
    const half_t* zeros;
    int nbm, nbn;
    int colmajor;            // tile order inside an XCD's contiguous chunk: 0 = column tile fastest (A panel stays in L2), 1 = row tile fastest (W panel stays in L2)
    unsigned* stamps;        // diagnostic builds (-DSDE_DIAG): per-wave section cycle sums of one block, else null
    int splitk; float* ws;   // splitk > 1: raw fp32 partial tiles go to ws[ks][M][Npad], reduced by splitk_reduce_kernel
    int ws_npp;              // gemm_ws.hip: panels per pass
};

// erf-form GELU, 0.5*x*(1+erf(x/sqrt2)), written as  max(x, 0) - |x| w(|x|),  w = 0.5 erfc(|x|/sqrt2) = (b1 t + b2 t^2 + b3 t^3) exp(-x^2/2),
// t = 1 / (1 + p |x| / sqrt2)  (Abramowitz & Stegun 7.1.25, |erf error| <= 2.5e-5; the coefficients below carry the factor 0.5).
// In this form there is no "1 - tail" and no copysign: nine plain VALU instructions and two transcendentals (v_rcp_f32, v_exp_f32 with
// log2(e) folded into its argument) = ~52 issue cycles per value, against ~72 for the five-term 7.1.26 form used until round 3 --
// the GEGLU projections are bound by exactly this epilogue.  Error against the exact function: <= 2.6e-5 absolute,
// 9e-6 rms over [-8, 8], i.e. 1.5 % of the fp16 rounding of the result it feeds (6e-4 rms).
__device__ __forceinline__ float gelu_erf(float x)
{
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.33267252f, ax, 1.0f));            // p / sqrt2, p = 0.47047
    float poly = fmaf(0.3739278f, t, -0.0479399f);                                 // 0.5 a3, 0.5 a2
    poly = fmaf(poly, t, 0.1740121f);                                              // 0.5 a1
    const float zl = ax * 0.84932180f;                                             // sqrt(log2(e) / 2): zl^2 = (x^2 / 2) log2(e)
    const float w = poly * t * __builtin_amdgcn_exp2f(-(zl * zl));
    return fmaf(-ax, w, fmaxf(x, 0.0f));
}

// com.microsoft.FastGelu (what scripts/optimize.py --approximate-gelu leaves in the graph): x * (0.5 + 0.5 * tanh(x * (B + C x^2)))
__device__ __forceinline__ float gelu_tanh_form(float x)
{
    const float u = x * (0.7978845608028654f + 0.035677408136300125f * x * x);
    const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));      // tanh(u); exp overflow -> 1, underflow -> -1
    return x * (0.5f + 0.5f * th);
}
__device__ __forceinline__ float gelu_sel(float x, int tanh_form) { return tanh_form ? gelu_tanh_form(x) : gelu_erf(x); }

// ------------------------------------------------------------------------------------------ epilogue (shared)
// Handles 4 consecutive columns n..n+3 of row m held in v[0..3].
__device__ __forceinline__ void epilogue4(const GemmDev& p, int m, int n, floatx4 v, floatx4 vg, long long zD, long long zR)
{
    if (m >= p.M) return;
    if (p.epi == EPI_STD) {
        if (n >= p.N) return;
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = v[r] * p.alpha;
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (n + r < p.N) o[r] += p.bias[n + r];
        }
        if (n + 3 < p.N) {
            if (p.R) {
                half4 rr = *(const half4*)(p.R + zR + (long long)m * p.ldr + n);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += (float)rr[r];
                if (p.Rlo) {
                    half4 rl = *(const half4*)(p.Rlo + zR + (long long)m * p.ldr + n);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] += (float)rl[r];
                }
            }
            half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
            *(half4*)(p.D + zD + (long long)m * p.ldd + n) = h;
            if (p.Dlo) {
                half4 l = {(half_t)(o[0] - (float)h[0]), (half_t)(o[1] - (float)h[1]), (half_t)(o[2] - (float)h[2]), (half_t)(o[3] - (float)h[3])};
                *(half4*)(p.Dlo + zD + (long long)m * p.ldd + n) = l;
            }
        } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) {
                float x = o[r];
                if (p.R) x += (float)p.R[zR + (long long)m * p.ldr + n + r];
                if (p.R && p.Rlo) x += (float)p.Rlo[zR + (long long)m * p.ldr + n + r];
                const half_t h = (half_t)x;
                p.D[zD + (long long)m * p.ldd + n + r] = h;
                if (p.Dlo) p.Dlo[zD + (long long)m * p.ldd + n + r] = (half_t)(x - (float)h);
            }
        }
    } else if (p.epi == EPI_GEGLU || p.epi == EPI_GEGLU_PAIR) {
        // v = value columns, vg = gate columns for output columns j..j+3 (j passed in n)
        int half_n = p.N >> 1;
        if (n >= half_n) return;
        half4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = v[r], g = vg[r];
            if (p.bias && n + r < half_n) { a += p.bias[n + r]; g += p.bias[half_n + n + r]; }
            h[r] = (half_t)(a * gelu_sel(g, p.gelu_tanh));
        }
        if (n + 3 < half_n) *(half4*)(p.D + zD + (long long)m * p.ldd + n) = h;
        else for (int r = 0; r < 4 && n + r < half_n; ++r) p.D[zD + (long long)m * p.ldd + n + r] = h[r];
    } else if (p.epi == EPI_NCHW_F32) {
        int hw = p.Ho * p.Wo;
        int nb = m / hw, pix = m - nb * hw;
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
            float x = v[r] + (p.bias ? p.bias[n + r] : 0.f);
            p.Df[zD + ((long long)nb * p.N + n + r) * hw + pix] = x;
        }
    } else if (p.epi == EPI_IMAGE) {
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
            float x = v[r] + (p.bias ? p.bias[n + r] : 0.f);
            x = x / 2.0f + 0.5f;
            x = fminf(fmaxf(x, 0.0f), 1.0f);
            p.Df[zD + (long long)m * p.N + n + r] = x;
        }
    } else {  // EPI_TRANS
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
            float x = v[r] * p.alpha + (p.bias ? p.bias[n + r] : 0.f);
            p.D[zD + (long long)(n + r) * p.ldd + m] = (half_t)x;
        }
    }
}

