// Micro-benchmark: do MFMA and transcendental VALU work from DIFFERENT waves of one SIMD overlap on gfx950?
// Each wave loops over [16 x v_mfma_f32_16x16x32_f16] and / or [32 x v_exp_f32 + 16 x v_cvt_pk_f16_f32] (the per-tile mix of the
// d=40 attention kernel).  Reported: SIMD cycles per wave-iteration for 1, 2, 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

template <int MODE>   // 1 = MFMA only, 2 = VALU only, 3 = both (MFMA block then VALU block), 4 = both, interleaved 1 MFMA : 2 exp : 1 cvt
__global__ void k(float* out, int iters)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 1e-3f + j); b[j] = (_Float16)(j * 0.25f); }
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = floatx4{0, 0, 0, 0};
    float e[32];
    for (int i = 0; i < 32; ++i) e[i] = threadIdx.x * 1e-4f - i;
    half2_t pk[16];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 7], 0, 0, 0);
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(e[2 * i]), "v"(e[2 * i + 1]));
        }
        if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i & 7], 0, 0, 0);
                asm volatile("v_exp_f32 %0, %0" : "+v"(e[2 * i]));
                asm volatile("v_exp_f32 %0, %0" : "+v"(e[2 * i + 1]));
                asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(e[2 * i]), "v"(e[2 * i + 1]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 32; ++i) s += e[i];
    if (MODE != 1) for (int i = 0; i < 16; ++i) s += (float)pk[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, int waves_per_simd, double ghz)
{
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, threads = 256 * waves_per_simd;
    k<MODE><<<256, threads>>>(d, 100);
    hipEventRecord(e0); k<MODE><<<256, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_wave_iter = ms * 1e6 / iters / waves_per_simd;     // SIMD time per wave-iteration
    printf("%-22s %d waves/SIMD: %7.1f ns per wave-iteration per SIMD (~%.0f cycles @%.2f GHz)\n", name, waves_per_simd, ns_per_wave_iter, ns_per_wave_iter * ghz, ghz);
    hipFree(d);
}
int main()
{
    const double ghz = 2.1;
    for (int w : {1, 2, 3, 4}) {
        run<1>("16 MFMA", w, ghz); run<2>("32 exp + 16 cvt", w, ghz); run<3>("both, sequential", w, ghz); run<4>("both, interleaved", w, ghz);
    }
    return 0;
}
