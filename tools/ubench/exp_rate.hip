// Micro-benchmark: issue rate of v_exp_f32 vs v_exp_f16 (and packed f16 fma) on gfx950, one wave per SIMD and 2 per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    _Float16 b0 = (_Float16)a0, b1 = (_Float16)a1, b2 = (_Float16)a2, b3 = (_Float16)a3, b4 = (_Float16)a4, b5 = (_Float16)a5, b6 = (_Float16)a6, b7 = (_Float16)a7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 1) {
            asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7));
        } else {
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7);
}
template <int MODE> void run(const char* name, int threads)
{
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    k<MODE><<<256, threads>>>(d, 100);
    hipEventRecord(e0); k<MODE><<<256, threads>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: waves = threads/256 ; instr per wave = iters*8
    double ns_per_instr = ms * 1e6 / (double(iters) * 8 * (threads / 256));
    printf("%-10s %4d threads/block: %.2f ns per wave-instruction per SIMD (~%.1f cycles @2.1GHz)\n", name, threads, ns_per_instr, ns_per_instr * 2.1);
    hipFree(d);
}
int main()
{
    for (int t : {256, 512, 1024}) { run<0>("v_exp_f32", t); run<1>("v_exp_f16", t); run<2>("v_fma_f32", t); }
    return 0;
}
