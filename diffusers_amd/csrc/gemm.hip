// gemm.hip — fp16 GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
//   D[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// A is an NHWC activation tensor read through an implicit im2col: for taps=9 the K axis runs over
// (ky,kx,cin) of a 3x3/pad-1 convolution (stride 1 or 2, optional on-the-fly nearest 2x upsample,
// optional channel concat of two sources); for taps=1 it is a plain row-major matrix.
// W is [Npad][K] fp16 (OHWI order), prepared once by the loader.
//
// MFMA path: 256 threads = 4 waves (2x2), block tile BM x BN x 64, v_mfma_f32_16x16x32_f16,
// both operand tiles staged with global_load_lds_dwordx4 (16 B/lane, LDS image lane-linear, XOR swizzle
// applied on the per-lane SOURCE address and again on the ds_read_b128 address), two LDS stages,
// one barrier per K-step, XCD-aware block remap.  Roofline: MFMA-bound (2*M*N*K flop).
// The MFMA is issued as D^T = W * A^T so each lane ends up with 4 consecutive n for one m (8-byte stores).
//
// Plain path: one thread per output element, same index maps, used for shapes the MFMA path does not take
// (K-chunk not a multiple of 64, tiny test models) and as an on-device cross-check.
#include "kernels.h"

#include <algorithm>

#include <mutex>
#include <set>
#include <utility>

#include "gemm_common.h"

// ------------------------------------------------------------------------------------------ MFMA kernel
template <int BM, int BN, bool CONV>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmDev p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    constexpr int AP = BM / 32, BP = BN / 32;  // glds passes per thread (32 rows of 128 B per pass)
    constexpr int MTW = BM / 32, NTW = BN / 32;  // 16-wide tiles per wave (wave tile = BM/2 x BN/2)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int z = blockIdx.z;

    // XCD-aware bijective remap: blocks b, b+8, b+16.. share an XCD (L2); give each XCD a contiguous tile range.
    const int nwg = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // each XCD re-reads whatever its tiles share: keep the larger operand's panel behind one L2 (host picks, launch_gemm)
    const int tm = p.colmajor ? bid % p.nbm : bid / p.nbn, tn = p.colmajor ? bid / p.nbm : bid - (bid / p.nbn) * p.nbn;
    const int m0 = tm * BM, n0 = tn * BN;

    const half_t* A0 = p.A0 + z * p.sA;
    const half_t* A1 = p.A1 ? p.A1 + z * p.sA : nullptr;
    const half_t* Wp = p.W + z * p.sW;

    // per-thread staging geometry
    const int srow = tid >> 3;                       // row inside a 32-row pass
    const int lchunk = (tid & 7) ^ (srow & 7);       // logical 16-B chunk this lane fetches (XOR swizzle on the source)
    int a_base[AP];                                  // conv: pixel index of image origin ; linear: row index
    int a_y[AP], a_x[AP];                            // conv: oy*stride-1, ox*stride-1
    bool a_ok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        int m = m0 + i * 32 + srow;
        a_ok[i] = m < p.M;
        if (CONV) {
            int hw = p.Ho * p.Wo;
            int nb = m / hw, rem = m - nb * hw;
            int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_base[i] = nb * p.Hs * p.Ws;
            a_y[i] = oy * p.stride + p.coff - 1;
            a_x[i] = ox * p.stride + p.coff - 1;
        } else {
            a_base[i] = m; a_y[i] = 0; a_x[i] = 0;
        }
    }
    const half_t* w_ptr[BP];
#pragma unroll
    for (int i = 0; i < BP; ++i) w_ptr[i] = Wp + (long long)(n0 + i * 32 + srow) * p.K + lchunk * 8;

    const int nk_all = p.K >> 6;
    const int ks = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * ks / p.splitk);
    const int nk = (int)((long long)nk_all * (ks + 1) / p.splitk) - kt0;

    auto stage = [&](int kt, int buf) {
        char* sa = smem + buf * STAGE;
        char* sb = sa + A_BYTES;
        const int k0 = (kt0 + kt) << 6;
        int tap = 0, c0 = k0;
        if (CONV) { const int kt_abs = kt0 + kt; const int chunk = kt_abs / 9; tap = kt_abs - chunk * 9; c0 = chunk << 6; }   // chunk-major K (kernels.h)
        const half_t* src = A0; int ld = p.lda0; int cc = c0;
        if (c0 >= p.C0) { src = A1; ld = p.lda1; cc = c0 - p.C0; }
        int dy = 0, dx = 0;
        if (CONV) { dy = tap / 3; dx = tap - dy * 3; }
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const half_t* g;
            if (CONV) {
                int iy = a_y[i] + dy, ix = a_x[i] + dx;
                bool ok = a_ok[i] && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                long long pix = a_base[i] + (iy >> p.ups) * p.Ws + (ix >> p.ups);
                g = ok ? src + pix * ld + cc + lchunk * 8 : p.zeros;
            } else {
                g = a_ok[i] ? src + (long long)a_base[i] * ld + cc + lchunk * 8 : p.zeros;
            }
            __builtin_amdgcn_global_load_lds(GLB_PTR(g), LDS_PTR(sa + (i * 256 + wid * 64) * 16), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < BP; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(w_ptr[i] + k0), LDS_PTR(sb + (i * 256 + wid * 64) * 16), 16, 0, 0);
    };

    floatx4 acc[NTW][MTW];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < MTW; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
        const char* sa = smem + buf * STAGE;
        const char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            half8 wf[NTW], af[MTW];
            const int q = kk * 4 + fg;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                int row = wn * (BN / 2) + nt * 16 + fr;
                wf[nt] = *(const half8*)(sb + (row * 8 + (q ^ (row & 7))) * 16);
            }
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                int row = wm * (BM / 2) + mt * 16 + fr;
                af[mt] = *(const half8*)(sa + (row * 8 + (q ^ (row & 7))) * 16);
            }
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < MTW; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[nt], af[mt], acc[nt][mt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // epilogue: acc[nt][mt][r] <-> n = n0 + wn*BN/2 + nt*16 + 4*fg + r ; m = m0 + wm*BM/2 + mt*16 + fr
    const long long zD = z * p.sD, zR = z * p.sR;
    if (p.splitk > 1) {
        float* ws = p.ws + (long long)ks * p.M * p.Npad;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int m = m0 + wm * (BM / 2) + mt * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                int n = n0 + wn * (BN / 2) + nt * 16 + 4 * fg;
                *(floatx4*)(ws + (long long)m * p.Npad + n) = acc[nt][mt];
            }
        }
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        const int m = m0 + wm * (BM / 2) + mt * 16 + fr;
        if (p.epi == EPI_GEGLU) {
#pragma unroll
            for (int nt = 0; nt + 1 < NTW; nt += 2) {
                int ncol = n0 + wn * (BN / 2) + nt * 16;      // interleaved column of the value block
                int j = (ncol >> 1) + 4 * fg;                 // output column
                epilogue4(p, m, j, acc[nt][mt], acc[nt + 1][mt], zD, zR);
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                int n = n0 + wn * (BN / 2) + nt * 16 + 4 * fg;
                epilogue4(p, m, n, acc[nt][mt], acc[nt][mt], zD, zR);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ split-K reduction + epilogue
__global__ void splitk_reduce_kernel(GemmDev p)
{
    const int groups = p.Npad >> 2;
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)p.M * groups) return;
    const int m = (int)(gid / groups), n = (int)(gid - (long long)m * groups) * 4;
    const float* ws = p.ws + (long long)m * p.Npad + n;
    const long long slab = (long long)p.M * p.Npad;
    floatx4 v = *(const floatx4*)ws;
    for (int k = 1; k < p.splitk; ++k) v += *(const floatx4*)(ws + k * slab);
    epilogue4(p, m, n, v, v, 0, 0);
}

// ------------------------------------------------------------------------------------------ plain kernel
__global__ void gemm_naive_kernel(GemmDev p)
{
    const int z = blockIdx.z;
    const half_t* A0 = p.A0 + z * p.sA;
    const half_t* A1 = p.A1 ? p.A1 + z * p.sA : nullptr;
    const half_t* Wp = p.W + z * p.sW;
    const bool gg = p.epi == EPI_GEGLU || p.epi == EPI_GEGLU_PAIR;
    const int Nout = gg ? (p.N >> 1) : p.N;
    const int groups = (Nout + 3) >> 2;
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)p.M * groups) return;
    const int m = (int)(gid / groups), n4 = (int)(gid - (long long)m * groups) * 4;
    const int Ctot = p.C0 + p.C1;
    int nb = 0, oy = 0, ox = 0;
    if (p.taps == 9) {
        int hw = p.Ho * p.Wo;
        nb = m / hw; int rem = m - nb * hw; oy = rem / p.Wo; ox = rem - oy * p.Wo;
    }
    floatx4 v = {0, 0, 0, 0}, vg = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) {
        int n = n4 + r;
        if (n >= Nout) break;
        // weight rows: GEGLU weights are interleaved in 16-blocks: out col j -> value row 32*(j/16)+j%16, gate row +16
        int wrow = n, grow = 0;
        if (p.epi == EPI_GEGLU) { wrow = 32 * (n >> 4) + (n & 15); grow = wrow + 16; }
        if (p.epi == EPI_GEGLU_PAIR) { wrow = 4 * (n >> 1) + (n & 1); grow = wrow + 2; }   // rows 4q..4q+3 = value_2q, value_2q+1, gate_2q, gate_2q+1
        float s = 0.f, sg = 0.f;
        for (int tap = 0; tap < p.taps; ++tap) {
            long long pix; bool ok = true;
            if (p.taps == 9) {
                int iy = oy * p.stride + p.coff - 1 + tap / 3, ix = ox * p.stride + p.coff - 1 + tap % 3;
                ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                pix = (long long)nb * p.Hs * p.Ws + (iy >> p.ups) * p.Ws + (ix >> p.ups);
            } else pix = m;
            if (!ok) continue;
            const half_t* wr = Wp + (long long)wrow * p.K;
            const half_t* wg = Wp + (long long)grow * p.K;
            const int ck = (p.taps > 1 && Ctot % 64 == 0) ? 64 : Ctot;   // K order: kernels.h
            const half_t* a0 = A0 + pix * p.lda0;
            for (int c = 0; c < p.C0; ++c) {
                const int kc = ((c / ck) * p.taps + tap) * ck + c % ck;
                float x = (float)a0[c];
                s += x * (float)wr[kc];
                if (gg) sg += x * (float)wg[kc];
            }
            if (p.C1) {
                const half_t* a1 = A1 + pix * p.lda1;
                for (int c = 0; c < p.C1; ++c) {
                    const int cc = p.C0 + c, kc = ((cc / ck) * p.taps + tap) * ck + cc % ck;
                    float x = (float)a1[c];
                    s += x * (float)wr[kc];
                    if (gg) sg += x * (float)wg[kc];
                }
            }
        }
        v[r] = s; vg[r] = sg;
    }
    epilogue4(p, m, n4, v, vg, z * p.sD, z * p.sR);
}

// ------------------------------------------------------------------------------------------ host side
struct GemmDev;
bool gemm_big_supported(const GemmArgs& a);                                        // gemm_big.hip
int gemm_big_bn(const GemmArgs& a);
bool gemm_mid_supported(const GemmArgs& a);                                        // gemm_mid.hip
void launch_gemm_mid(const LaunchCtx& ctx, GemmDev& p, int batch, bool conv);
void launch_gemm_mid128(const LaunchCtx& ctx, GemmDev& p, int batch);
bool gemm_midh_supported(const GemmArgs& a);
void launch_gemm_midh(const LaunchCtx& ctx, GemmDev& p);
bool gemm_ws_supported(const GemmArgs& a);
void launch_gemm_ws(const LaunchCtx& ctx, GemmDev& p);
void launch_gemm_big(const LaunchCtx& ctx, GemmDev& p, int batch, bool conv);

int gemm_npad(int N) { return (N % 160 == 0 || N % 128 == 0) ? N : round_up(N, 128); }

static GemmDev to_dev(const LaunchCtx& ctx, const GemmArgs& a)
{
    GemmDev p;
    p.A0 = a.A0; p.A1 = a.A1; p.C0 = a.C0; p.C1 = a.C1; p.lda0 = a.lda0; p.lda1 = a.lda1;
    p.M = a.M; p.N = a.N; p.K = a.K; p.taps = a.taps; p.stride = a.stride; p.coff = a.coff; p.ups = a.ups;
    p.Hi = a.Hi; p.Wi = a.Wi; p.Ho = a.Ho; p.Wo = a.Wo;
    p.Hs = a.ups ? (a.Hi + 1) / 2 : a.Hi; p.Ws = a.ups ? (a.Wi + 1) / 2 : a.Wi;   // ups: virtual (Hi x Wi) = nearest-2x of (Hs x Ws), cropped when odd
    p.W = a.W; p.Npad = a.Npad; p.bias = a.bias; p.R = a.R; p.Rlo = a.Rlo; p.Dlo = a.Dlo; p.ldr = a.ldr; p.D = a.D; p.Df = a.Df; p.ldd = a.ldd;
    p.epi = a.epi; p.gelu_tanh = a.gelu_tanh; p.alpha = a.alpha; p.sA = a.sA; p.sW = a.sW; p.sD = a.sD; p.sR = a.sR; p.zeros = ctx.zeros;
    p.nbm = p.nbn = 0; p.splitk = 1; p.ws = nullptr; p.stamps = nullptr; p.ws_npp = 1;
    // fabric reads ~ A + 8*W with column tiles fastest, 8*A + W with row tiles fastest (8 XCDs, each with its own L2)
    {
        static const int cm_env = getenv("SDE_COLMAJOR") ? atoi(getenv("SDE_COLMAJOR")) : -1;   // tuning hook
        const double bytesA = (double)a.M * (a.C0 + a.C1) * 2.0 / (a.stride * a.stride) * (a.ups ? 0.25 : 1.0), bytesW = (double)a.Npad * a.K * 2.0;
        // Whichever operand is larger stays behind one L2.  In time only the 8x8 level gains (3-4 %) and near the break-even it is a wash,
        // but the fabric traffic follows the model: 16x16-level 3x3 convs (W = 29.5 / 59 MB, A = 10.5 / 21 MB) fetch 185 MB per launch
        // with column tiles fastest and 85 MB with row tiles fastest (FETCH_SIZE, raw); the 32x32 level (A > W) 66 against 80 MB.
        static const double cm_ratio = getenv("SDE_COLMAJOR_RATIO") ? atof(getenv("SDE_COLMAJOR_RATIO")) : 1.0;   // tuning hook
        p.colmajor = cm_env >= 0 ? cm_env : (bytesW > cm_ratio * bytesA ? 1 : 0);
    }
    return p;
}

bool gemm_mfma_supported(const GemmArgs& a)
{
    const int Ctot = a.C0 + a.C1;
    if (a.K % 64 || a.K != a.taps * Ctot) return false;
    if (a.taps == 9 && (Ctot % 64 || a.C0 % 64)) return false;   // a 64-wide K chunk must stay inside one tap / one source
    if (a.taps == 1 && a.C1 && a.C0 % 64) return false;
    if (a.lda0 % 8 || (a.C1 && a.lda1 % 8)) return false;          // 16-B aligned source chunks
    if (a.Npad % 128 && a.Npad % 160) return false;
    if (a.epi == EPI_GEGLU && (a.N % 64 || a.Npad % 128)) return false;
    if (a.epi == EPI_GEGLU_PAIR) return false;   // in-fours layout: gemm_big.hip only
    if ((a.epi == EPI_STD || a.epi == EPI_GEGLU) && (a.ldd % 4 || (a.R && a.ldr % 4))) return false;
    if ((reinterpret_cast<uintptr_t>(a.A0) | reinterpret_cast<uintptr_t>(a.W)) & 15) return false;
    if ((a.sA | a.sW) % 8 || (a.sD | a.sR) % 4) return false;
    return true;
}

void allow_dynamic_lds(const void* kernel, int bytes)
{
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({dev, kernel})) return;
    HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({dev, kernel});
}

template <int BM, int BN, bool CONV>
static void launch_mfma(const LaunchCtx& ctx, GemmDev& p, int batch)
{
    constexpr int smem = 2 * (BM + BN) * 128;
    auto kern = gemm_mfma_kernel<BM, BN, CONV>;
    allow_dynamic_lds((const void*)kern, smem);
    p.nbm = cdiv(p.M, BM);
    p.nbn = p.Npad / BN;
    // split-K when the tile grid cannot fill the chip (2 blocks/CU x 256 CUs) and K is long: partial slabs + reduce kernel
    p.splitk = 1;
    const int tiles = p.nbm * p.nbn, nk = p.K >> 6;
    if (batch == 1 && p.epi != EPI_GEGLU && ctx.splitk_ws && tiles < 512 && nk >= 32) {
        int sk = std::min(std::min((512 + tiles / 2) / tiles, nk / 16), 8);   // one full wave of blocks (2 per CU)
        if (ctx.splitk_force > 0) sk = std::max(1, std::min(ctx.splitk_force, nk / 4));
        while (sk > 1 && (size_t)sk * p.M * p.Npad * sizeof(float) > ctx.splitk_ws_bytes) --sk;
        if (sk > 1) { p.splitk = sk; p.ws = ctx.splitk_ws; }
    }
    dim3 grid(p.nbm * p.nbn, p.splitk, batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, ctx.stream, p);   // p.splitk > 1: the caller follows up with launch_splitk_reduce
}

// split-K second pass (own profiler class: one class per kernel symbol)
static void launch_splitk_reduce(const LaunchCtx& ctx, const GemmDev& p)
{
    const long long total = (long long)p.M * (p.Npad >> 2);
    ProfScope ps(ctx, "splitk_reduce", 0.0, ((double)p.splitk * 4.0 + 2.0) * (double)p.M * p.Npad);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, ctx.stream, p);
}

// profiler class name; SDE_PROF_SHAPES=1 appends the problem shape (tools/shape_profile.py)
static const char* prof_name(const char* base, const GemmArgs& a, const char* kern)
{
    static const bool shapes = getenv("SDE_PROF_SHAPES") && atoi(getenv("SDE_PROF_SHAPES"));
    if (!shapes) return base;
    static thread_local char buf[48];
    snprintf(buf, sizeof(buf), "%s%s M%d N%d K%d%s%s", a.taps == 9 ? "c3" : ((a.epi == EPI_GEGLU || a.epi == EPI_GEGLU_PAIR) ? "gg" : "g1"), kern, a.M, a.N, a.K,
             a.R ? " R" : "", a.batch > 1 ? " b" : "");
    return buf;
}

void launch_gemm(const LaunchCtx& ctx, const GemmArgs& a)
{
    SDE_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0 && a.A0 && a.W, SDE_ERR_INVALID_ARGUMENT, "launch_gemm: empty problem");
    SDE_REQUIRE(a.K == a.taps * (a.C0 + a.C1), SDE_ERR_INVALID_ARGUMENT, "launch_gemm: K != taps*(C0+C1)");
    SDE_REQUIRE(a.N <= a.Npad, SDE_ERR_INVALID_ARGUMENT, "launch_gemm: N > Npad");
    SDE_REQUIRE((!a.Rlo && !a.Dlo) || a.epi == EPI_STD, SDE_ERR_INVALID_ARGUMENT, "launch_gemm: split-fp16 residual / result needs EPI_STD");
    SDE_REQUIRE(!a.Rlo || a.R, SDE_ERR_INVALID_ARGUMENT, "launch_gemm: Rlo without R");
    if (a.taps == 9)
        SDE_REQUIRE(a.M % (a.Ho * a.Wo) == 0 && a.Ho > 0 && a.Wo > 0 && a.Hi > 0 && a.Wi > 0, SDE_ERR_INVALID_ARGUMENT,
                    "launch_gemm: conv geometry");
    if (ctx.f32) { f32_launch_gemm(ctx, a); return; }
    GemmDev p = to_dev(ctx, a);
    const double flops = 2.0 * a.M * (double)a.N * a.K * a.batch;
    const double bytes = ((double)a.M * a.K / (a.taps == 9 ? 9.0 : 1.0) + (double)a.N * a.K + (double)a.M * a.N * (1 + (a.R ? 1 : 0) + (a.Rlo ? 1 : 0) + (a.Dlo ? 1 : 0))) * 2.0 * a.batch;
    // 3x3 stride-1 convs whose width is a multiple of 128 but not of 160 (the VAE decoder: 128 / 256 / 512 channels): the halo kernel on
    // 256x128 tiles.  Measured at batch 8 against the 256x256 / 4-wave kernels they ran on (us): 512^2 128->128 1,150 -> 700;
    // 256^2 256->256 621 -> 530; 128^2 512->512 575 -> 483, with fused upsample 2,444 -> 1,870; 64^2 512->512 146 -> 124.
    static const int halo128_env = getenv("SDE_HALO128") ? atoi(getenv("SDE_HALO128")) : 1;   // tuning hook
    if (!ctx.force_naive && !ctx.force_big && !ctx.force_mid && a.taps == 9 && a.Npad % 160 != 0 && !a.Rlo && !a.Dlo &&
        (halo128_env || ctx.force_midh) && gemm_midh_supported(a) && (ctx.force_midh || cdiv(a.M, 256) * (a.Npad / 128) >= 192)) {
        {
            ProfScope ps(ctx, prof_name("conv3x3_mfma_halo", a, "Q"), flops, bytes);
            launch_gemm_midh(ctx, p);
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    // K = 320 projections over many rows (the 64x64 UNet level): the W-stationary streaming kernel (gemm_ws.hip).  Measured against the
    // 256x320 kernel at M = 65536 (us, HIP events): GEGLU N = 2560 150.7 / 163.8, q / k / v N = 960 66.5 / 70.8, N = 320 29.2 / 28.6 (a tie:
    // both sit ~20 % above what a LayerNorm pass over the same 84 MB takes, and stay where they are)
    static const int ws_env = getenv("SDE_WS") ? atoi(getenv("SDE_WS")) : 1;   // tuning hook: 0 never, 1 heuristic, 2 whenever supported
    static const int ws_min_m = getenv("SDE_WS_MIN_M") ? atoi(getenv("SDE_WS_MIN_M")) : 16384;
    if (!ctx.force_naive && !ctx.force_big && !ctx.force_mid && !ctx.force_mid128 && !ctx.force_midh && gemm_ws_supported(a) &&
        (ctx.force_ws || ws_env == 2 || (ws_env && a.M >= ws_min_m && (a.epi == EPI_GEGLU_PAIR || a.Npad >= 960)))) {
        {
            ProfScope ps(ctx, prof_name(a.epi == EPI_GEGLU_PAIR ? "gemm_geglu_mfma_ws" : "gemm_mfma_ws", a, "W"), flops, bytes);
            launch_gemm_ws(ctx, p);
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    static const int big_env = getenv("SDE_BIG") ? atoi(getenv("SDE_BIG")) : 1;   // tuning hook: 0 never, 1 heuristic, 2 whenever supported
    if (!ctx.force_naive && big_env && !ctx.force_mid && !ctx.force_midh && !ctx.force_mid128 && gemm_big_supported(a)) {
        // 8-wave 256x320 tiles, one block per CU: worth it once the grid (with split-K on long K) covers most of the chip
        const int tiles = cdiv(a.M, 256) * (a.Npad / gemm_big_bn(a)) * a.batch, nk = a.K >> 6;
        int sk = 1;
        if (a.batch == 1 && ctx.splitk_ws && tiles < 192 && nk >= 64) {
            sk = std::min(std::min((256 + tiles / 2) / tiles, nk / 32), 2);   // deeper splits: the 4-wave kernel's slabs are cheaper
            if (ctx.splitk_force > 0) sk = std::max(1, std::min(ctx.splitk_force, nk / 4));
            while (sk > 1 && (size_t)sk * a.M * a.Npad * sizeof(float) > std::min(ctx.splitk_ws_bytes, (size_t)48 << 20)) --sk;   // slab traffic must stay small next to the GEMM
        }
        // 3x3 convs the halo kernel takes (64x64 level, N = 320 as two 160-column tiles): 185 us against 204 at K = 5760, 102.6 against
        // 110.6 at K = 2880 (a tie, 111.5 / 111.0, until the halo kernel's epilogue stopped spilling)
        static const int halo64_env = getenv("SDE_HALO64") ? atoi(getenv("SDE_HALO64")) : 1;   // tuning hook
        const bool to_halo = halo64_env && !ctx.force_big && big_env != 2 && a.taps == 9 && a.K >= 2880 && sk == 1 && a.Npad % 160 == 0 && gemm_midh_supported(a) &&
                             cdiv(a.M, 256) * (a.Npad / 160) >= 192;
        if (!to_halo && (big_env == 2 || ctx.force_big || tiles * sk >= 192 || a.epi == EPI_GEGLU_PAIR)) {
            p.splitk = sk; p.ws = sk > 1 ? ctx.splitk_ws : nullptr;
            {
                ProfScope ps(ctx, prof_name(a.taps == 9 ? "conv3x3_mfma_256x320" : (a.epi == EPI_GEGLU_PAIR ? "gemm_geglu_mfma_256x320" : "gemm_mfma_256x320"), a, "B"), flops, bytes);
                launch_gemm_big(ctx, p, a.batch, a.taps == 9);
            }
            if (sk > 1) launch_splitk_reduce(ctx, p);
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    static const int mid_env = getenv("SDE_MID") ? atoi(getenv("SDE_MID")) : 1;   // tuning hook: 0 never, 1 heuristic, 2 whenever supported
    static const int mid128_env = getenv("SDE_MID128") ? atoi(getenv("SDE_MID128")) : 1;   // tuning hook: 0 never, 1 heuristic
    if (!ctx.force_naive && !ctx.force_big && !ctx.force_mid && a.taps == 1 && gemm_mid_supported(a)) {
        // linears whose 256-row tiles leave the chip half empty (the 16x16 level, M = 4096 at batch 16): 128x160 tiles, loader waves,
        // loads three K-tiles ahead.  The 4-wave 64x160 kernel they ran on moves (1/64 + 1/160) / (1/128 + 1/160) = 1.56x the bytes from
        // L2 into LDS, and that path (~70 GB/s per CU, MI355X_MICROARCH.md) is what bounds these launches.
        const int tiles256 = cdiv(a.M, 256) * (a.Npad / 160) * a.batch, tiles128 = cdiv(a.M, 128) * (a.Npad / 160) * a.batch;
        // measured (M = 4096, N = 1280, us): K = 1280 27.9 -> 24.5, K = 2560 41.0 -> 38.1; K = 5120 stays on the 256x160 kernel with a
        // 2-way split (58.3 against 65.7 here)
        if (ctx.force_mid128 || (mid128_env && tiles256 < 192 && tiles128 >= 192 && (a.K >> 6) < 64)) {
            {
                ProfScope ps(ctx, prof_name("gemm_mfma_ldw128", a, "H"), flops, bytes);
                launch_gemm_mid128(ctx, p, a.batch);
            }
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    if (!ctx.force_naive && mid_env && !ctx.force_big && gemm_mid_supported(a)) {
        // 8-wave 256x160 tiles with a 3-deep ring: the 32x32 / 16x16 UNet levels (one block per CU)
        const int tiles = cdiv(a.M, 256) * (a.Npad / 160) * a.batch, nk = a.K >> 6;
        int sk = 1;
        if (a.batch == 1 && ctx.splitk_ws && tiles < 192 && nk >= 64) {
            // as many ways as fill one block per CU: 16x16 level (128 tiles) 2 ways; the 640-channel stride-2 downsample into 16x16
            // (64 tiles) 4 ways; 8x8 level (32 tiles, K = 11520 / 23040) 8 ways: 47.6 / 70.1 us against 54.7 / 94.4 us on the
            // 4-wave kernel (MI355X, UNet batch 16)
            sk = std::min(std::min((256 + tiles / 2) / tiles, nk / 16), 8);
            if (ctx.splitk_force > 0) sk = std::max(1, std::min(ctx.splitk_force, nk / 4));
            while (sk > 1 && (size_t)sk * a.M * a.Npad * sizeof(float) > std::min(ctx.splitk_ws_bytes, (size_t)48 << 20)) --sk;
        }
        // 3x3 stride-1 convolutions over whole 16 x 16 patches: the halo kernel (input staged once per 64-channel chunk, not per tap)
        static const int midh_env = getenv("SDE_MIDH") ? atoi(getenv("SDE_MIDH")) : 1;   // tuning hook: 0 never, 1 whenever supported
        const bool halo = !ctx.force_mid && (midh_env || ctx.force_midh) && gemm_midh_supported(a);
        if (halo && (mid_env == 2 || ctx.force_midh || tiles * sk >= 192)) {
            sk = std::min(sk, a.K / 576);
            p.splitk = sk; p.ws = sk > 1 ? ctx.splitk_ws : nullptr;
            {
                ProfScope ps(ctx, prof_name("conv3x3_mfma_halo", a, "P"), flops, bytes);
                launch_gemm_midh(ctx, p);
            }
            if (sk > 1) launch_splitk_reduce(ctx, p);
            HIP_CHECK(hipGetLastError());
            return;
        }
        if (mid_env == 2 || ctx.force_mid || tiles * sk >= 192) {
            p.splitk = sk; p.ws = sk > 1 ? ctx.splitk_ws : nullptr;
            {
            ProfScope ps(ctx, prof_name(a.taps == 9 ? "conv3x3_mfma_256x160" : "gemm_mfma_256x160", a, "M"), flops, bytes);
#ifdef SDE_DIAG
            static const bool stamps = getenv("SDE_STAMPS") && atoi(getenv("SDE_STAMPS"));
            if (stamps && sk == 1) p.stamps = (unsigned*)ctx.splitk_ws;
#endif
            launch_gemm_mid(ctx, p, a.batch, a.taps == 9);
#ifdef SDE_DIAG
            if (p.stamps) {
                unsigned h[67];
                HIP_CHECK(hipStreamSynchronize(ctx.stream));
                HIP_CHECK(hipMemcpy(h, p.stamps, sizeof(h), hipMemcpyDeviceToHost));
                fprintf(stderr, "[stamps] loop: %u shader cycles in %u x 10 ns = %.3f GHz, %.1f cycles per K-tile\n", h[65], h[66], h[65] / (h[66] * 10.0), (double)h[65] / h[64]);
                fprintf(stderr, "[stamps] M=%d N=%d K=%d: cycles per K-tile and wave: L0 | wait B | M0 | wait B | L1(+vmcnt g1) | wait B | M1(+vmcnt g0) | wait B\n", a.M, a.N, a.K);
                for (int w = 0; w < 8; ++w) {
                    fprintf(stderr, "  wave %d:", w);
                    for (int i = 0; i < 8; ++i) fprintf(stderr, " %7.1f", (double)h[w * 8 + i] / h[64]);
                    fprintf(stderr, "\n");
                }
            }
#endif
            }
            if (sk > 1) launch_splitk_reduce(ctx, p);
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    if (!ctx.force_naive && gemm_mfma_supported(a)) {
        {
        ProfScope ps(ctx, prof_name(a.taps == 9 ? "conv3x3_mfma" : (a.epi == EPI_GEGLU ? "gemm_geglu_mfma" : "gemm_mfma"), a, "S"), flops, bytes);
        // tile height: 128 rows unless that leaves the chip (256 CUs x 2 blocks) under-filled and split-K (<= 4 ways,
        // long K only) cannot make up for it -> 64 rows.  Measured on MI355X at UNet batch 16 (tools/microbench.py).
        static const int bm64_env = getenv("SDE_BM64") ? atoi(getenv("SDE_BM64")) : -1;   // tuning hook
        const int nbn_ = a.Npad / ((a.Npad % 160 == 0 && a.epi != EPI_GEGLU) ? 160 : 128);
        const int tiles128 = cdiv(a.M, 128) * nbn_ * a.batch;
        const bool can_split = a.batch == 1 && a.epi != EPI_GEGLU && ctx.splitk_ws && (a.K >> 6) >= 64;
        bool small_m = tiles128 < 512 && !(can_split && tiles128 >= 128);
        if (bm64_env >= 0) small_m = a.M <= bm64_env;
        const bool bn160 = (a.Npad % 160 == 0) && a.epi != EPI_GEGLU;
        if (a.taps == 9) {
            if (bn160) { if (small_m) launch_mfma<64, 160, true>(ctx, p, a.batch); else launch_mfma<128, 160, true>(ctx, p, a.batch); }
            else       { if (small_m) launch_mfma<64, 128, true>(ctx, p, a.batch); else launch_mfma<128, 128, true>(ctx, p, a.batch); }
        } else {
            if (bn160) { if (small_m) launch_mfma<64, 160, false>(ctx, p, a.batch); else launch_mfma<128, 160, false>(ctx, p, a.batch); }
            else       { if (small_m) launch_mfma<64, 128, false>(ctx, p, a.batch); else launch_mfma<128, 128, false>(ctx, p, a.batch); }
        }
        }
        if (p.splitk > 1) launch_splitk_reduce(ctx, p);
    } else {
        ProfScope ps(ctx, a.taps == 9 ? "conv3x3_plain" : "gemm_plain", flops, bytes);
        const int Nout = (a.epi == EPI_GEGLU || a.epi == EPI_GEGLU_PAIR) ? a.N / 2 : a.N;
        long long total = (long long)a.M * ((Nout + 3) / 4);
        dim3 grid((unsigned)cdiv64(total, 256), 1, a.batch);
        hipLaunchKernelGGL(gemm_naive_kernel, grid, dim3(256), 0, ctx.stream, p);
    }
    HIP_CHECK(hipGetLastError());
}
