// gemm_mid.hip — 8-wave 256x160x64 fp16 GEMM / implicit-GEMM convolution with a 3-deep LDS ring, for gfx950 (MI355X).
//
// Same contract as gemm.hip / gemm_big.hip.  It takes the shapes between the two: the 32x32 and 16x16 UNet levels
// (M = 16384 / 4096 at batch 16), whose grids cannot fill 256 CUs with 256x320 tiles but where the 4-wave 128x160 kernel
// keeps the global->LDS path ~90 % busy.  A 256x160 tile needs 65 % of that path and gives 256 blocks at M=16384, N=640.
//
// Structure:
//   * 512 threads = 8 waves as 4 (M) x 2 (N); wave tile 64 x 80 = 4 x 5 MFMA 16x16x32 tiles (80 accumulator VGPRs),
//     D^T = W * A^T as in the other kernels.
//   * LDS: 3 K-tile buffers x (256 + 160) rows x 128 B = 156 KiB, filled by buffer_load_dwordx4 ... lds (range-checked
//     padding, source-side XOR swizzle).  The ring is three deep because a K-tile is only two phases long here: loads run
//     two tiles ahead and are retired with a COUNTED s_waitcnt vmcnt(N) that leaves the newest tile in flight.
//   * Each K-tile is 2 phases (k-halves) of {9 ds_read_b128 + load issue | barrier | 20 MFMA | barrier}; waves 4-7 run one
//     barrier behind waves 0-3 (ping-pong on every SIMD, as in gemm_big.hip).
//   * Buffer reuse: tile t+2 goes into the buffer of tile t-1.  Waves 4-7 finish reading it one interval into tile t, so
//     waves 0-3 may only stage rows nobody else reads (their own A rows 0-127) in their first load section.
#include "gemm_common.h"

#include <algorithm>
#include <type_traits>
#include <utility>

namespace {
constexpr int BM = 256, BN = 160, WN = 80, NT = 5, MT = 4;
constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;       // 53,248 B per K-tile
constexpr int SMEM_BYTES = 3 * STAGE;                                // 159,744 B
constexpr unsigned OOB = 0x80000000u;
constexpr int RSRC_FLAGS = 0x00020000;
}  // namespace

// Issue schedule: how many of a wave's staging parts go into each section {L0, M0, L1, M1} of a K-tile, for waves 0-3 (7 parts)
// and waves 4-7 (6 parts).  Parts issued in an M section are spread among its 20 MFMAs.  Legal: waves 0-3 at most 2 parts in L0
// (see the header), every part issued before the wave's counted wait of the NEXT tile.  Entry 0 is the product schedule.
namespace {
constexpr int kMidSched[4][2][4] = {
    {{2, 1, 4, 0}, {3, 0, 3, 0}},
    {{2, 2, 2, 1}, {2, 1, 2, 1}},
    {{0, 4, 0, 3}, {0, 3, 0, 3}},
    {{1, 2, 2, 2}, {1, 2, 1, 2}},
};
constexpr int sched_base(int v, int g, int sec) { int b = 0; for (int i = 0; i < sec; ++i) b += kMidSched[v][g][i]; return b; }
constexpr int sched_slot(int k, int n) { return (k + 1) * 20 / (n + 1) - 1; }   // MFMA index after which part k of n is issued
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
}  // namespace

// epilogue shared by both kernels: acc[nt][mt][r] <-> n = n0 + wc*80 + nt*16 + 4*fg + r ; m = m0 + wr*64 + mt*16 + fr
#ifndef MID_RG
#define MID_RG(LO) ((LO) ? 1 : 2)
#endif
// LO: split-fp16 residual / result (hi + lo pairs); its own instance so the plain epilogue keeps its register budget (gemm_big.hip)
// PATCH: the tile's 256 rows are a 16 x 16 patch of output pixels (halo kernel below): tile row r is pixel m0 + (r >> 4) * Wo + (r & 15)
// NT_: 16-column tiles per wave (5: 160-column block tile; 4: 128-column, the halo kernel on VAE widths)
template <bool LO, int SLICE = SMEM_BYTES / 8, bool PATCH = false, int NT_ = 5, int RG_ = MID_RG(LO)>
__device__ __forceinline__ void mid_epilogue(const GemmDev& p, char* smem, floatx4 (&acc)[NT_][MT], int m0, int n0, int ks, int z,
                                             int wid, int wr, int wc, int lane, int fr, int fg)
{
    constexpr int NT = NT_, WN = NT_ * 16;   // shadow the file-scope constants
    // ---- epilogue: acc[nt][mt][r] <-> n = n0 + wc*80 + nt*16 + 4*fg + r ; m = m0 + wr*64 + mt*16 + fr
    const long long zD = z * p.sD, zR = z * p.sR;
    auto rowm = [&](int r) { return PATCH ? m0 + (r >> 4) * p.Wo + (r & 15) : m0 + r; };
    if (p.splitk > 1) {
        float* ws = p.ws + (long long)ks * p.M * p.Npad;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = rowm(wr * 64 + mt * 16 + fr);
            if (m < p.M) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int n = n0 + wc * WN + nt * 16 + 4 * fg;
                    *(floatx4*)(ws + (long long)m * p.Npad + n) = acc[nt][mt];
                }
            }
        }
        return;
    }
    // EPI_STD through the wave's own LDS slice (see gemm_big.hip): fp32 staging, 16-byte row-order stores
    char* const wlds = smem + wid * SLICE;
    const bool vec_ok = (p.ldd & 7) == 0;
    constexpr int PITCH = WN * 4 + 32, CH = WN / 8;
    constexpr int RG = RG_;                 // 16-row groups per step (gemm_big.hip: the split-fp16 instance takes one)
    constexpr int NCH = RG * 16 * CH, NI = (NCH + 63) / 64;
    // 16-row steps (RG_ == 1): the bias is re-read per step (L1 hits) instead of living in 20 registers next to the residual sets: the
    // 12-wave kernels have 168 VGPRs and every spilled dword in this epilogue is a scratch round trip in the tile's latency chain
    constexpr bool BIL = RG_ == 1;   // bias in the loop: the split-fp16 instances and the 12-wave kernels
    float bv[BIL ? 1 : NT][4];
    if constexpr (!BIL) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + wc * WN + nt * 16 + 4 * fg;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[nt][r] = (p.bias && n < p.N) ? p.bias[n + r] : 0.f;
        }
    }
    const half_t* R = p.R ? p.R + zR : nullptr;
    const half_t* Rlo = (LO && p.R && p.Rlo) ? p.Rlo + zR : nullptr;   // split-fp16 residual / result (kernels.h)
    half_t* D = p.D + zD;
    half_t* Dlo = (LO && p.Dlo) ? p.Dlo + zD : nullptr;
    const bool rvec_ok = R && (p.ldr & 7) == 0;
#pragma unroll
    for (int ms = 0; ms < MT / RG; ++ms) {
        half8 rr[NI], rl[LO ? NI : 1];
        if (R) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int id = min(lane + 64 * i, NCH - 1), row = id / CH, ch = id - row * CH;
                const int m = min(rowm(wr * 64 + ms * (16 * RG) + row), p.M - 1), n = min(n0 + wc * WN + ch * 8, p.N - 8);
                if (LO || rvec_ok) rr[i] = *(const half8*)(R + (long long)m * p.ldr + n);
                else for (int e = 0; e < 8; ++e) rr[i][e] = R[(long long)m * p.ldr + n + e];
                if constexpr (LO) if (Rlo) {
                    rl[i] = *(const half8*)(Rlo + (long long)m * p.ldr + n);
                }
            }
        }
#pragma unroll
        for (int mq = 0; mq < RG; ++mq) {
            const int mt = ms * RG + mq;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                floatx4 o;
                if constexpr (BIL) {
                    const int n = n0 + wc * WN + nt * 16 + 4 * fg;
#pragma unroll
                    for (int r = 0; r < 4; ++r) bv[0][r] = (p.bias && n < p.N) ? p.bias[n + r] : 0.f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = acc[nt][mt][r] * p.alpha + bv[BIL ? 0 : nt][r];
                *(floatx4*)(wlds + (mq * 16 + fr) * PITCH + (nt * 16 + 4 * fg) * 4) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int id = lane + 64 * i, idc = min(id, NCH - 1), row = idc / CH, ch = idc - row * CH;
            const int m = rowm(wr * 64 + ms * (16 * RG) + row), n = n0 + wc * WN + ch * 8;
            const floatx4 lo = *(const floatx4*)(wlds + row * PITCH + ch * 32), hi = *(const floatx4*)(wlds + row * PITCH + ch * 32 + 16);
            const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            half8 v, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float x = R ? f[e] + (float)rr[i][e] : f[e];
                if constexpr (LO) if (Rlo) x += (float)rl[i][e];
                v[e] = (half_t)x;
                if constexpr (LO) vl[e] = (half_t)(x - (float)v[e]);
            }
            if (id < NCH && m < p.M && n < p.N) {
                if (LO || (vec_ok && n + 8 <= p.N)) {   // LO: the host only sends 16-byte aligned rows and N % 8 == 0 here
                    *(half8*)(D + (long long)m * p.ldd + n) = v;
                    if constexpr (LO) if (Dlo) *(half8*)(Dlo + (long long)m * p.ldd + n) = vl;
                } else for (int e = 0; e < 8 && n + e < p.N; ++e) {
                    D[(long long)m * p.ldd + n + e] = v[e];
                    if constexpr (LO) if (Dlo) Dlo[(long long)m * p.ldd + n + e] = vl[e];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

// staging parts of one K-tile: 0-3 = A rows 64j..64j+63, 4-5 = W rows 0-63 / 64-127, 6 = W rows 128-159 (waves 0-3 only)
template <bool CONV, bool UPS, int VAR, bool LO = false>
__global__ __launch_bounds__(512) void gemm_mid_kernel(GemmDev p)
{
    static_assert(kMidSched[VAR][0][0] <= 2, "waves 0-3 may stage only their own A rows in L0");
    static_assert(sched_base(VAR, 0, 4) == 7 && sched_base(VAR, 1, 4) == 6, "every part exactly once");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1, grp = wid >> 2;
    const int z = blockIdx.z;
    const int fr = lane & 15, fg = lane >> 4;

    const int nwg = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = p.colmajor ? bid % p.nbm : bid / p.nbn, tn = p.colmajor ? bid / p.nbm : bid - (bid / p.nbn) * p.nbn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk_all = p.K >> 6;
    const int ks = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * ks / p.splitk);
    const int nk = (int)((long long)nk_all * (ks + 1) / p.splitk) - kt0;

    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A0 + z * p.sA), 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)((p.A1 ? p.A1 : p.A0) + z * p.sA), 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + z * p.sW), 0, OOB, RSRC_FLAGS);

    // ---- staging geometry (as gemm_big.hip): wave w stages rows 8w..8w+7 of every 64-row part
    const int srow = lane >> 3;
    const unsigned lchunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
    unsigned a_off0[4], a_off1[4], a_mask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + j * 64 + wid * 8 + srow;
        const bool ok = m < p.M;
        int pix = m;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int nb = m / hw, rem = m - nb * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int cy = oy * p.stride + p.coff, cx = ox * p.stride + p.coff;
            unsigned mask = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
                if (ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) mask |= 1u << t;
            }
            mask |= (unsigned)(cy & 1) << 9 | (unsigned)(cx & 1) << 10;
            a_mask[j] = mask;
            pix = nb * p.Hs * p.Ws + (cy >> (UPS ? 1 : 0)) * p.Ws + (cx >> (UPS ? 1 : 0));
        } else {
            a_mask[j] = ok ? 1u : 0u;
        }
        a_off0[j] = (unsigned)pix * (unsigned)(p.lda0 * 2) + lchunk16;
        a_off1[j] = (unsigned)pix * (unsigned)(p.lda1 * 2) + lchunk16;
    }
    const unsigned w_off = (unsigned)((n0 + wid * 8 + srow) * p.K) * 2u + lchunk16;
    const int w_part = 128 * p.K;

    // staging state of the K-tile currently being issued (K order: kernels.h)
    int st_k0 = kt0 << 6, st_tap = 0, st_c = st_k0;
    if (CONV) { st_tap = kt0 % 9; st_c = (kt0 / 9) << 6; }
    bool st_second = false;
    unsigned st_bit = 1u;
    int st_add = 0, st_ld2 = 0, st_ey0 = 0, st_ey1 = 0, st_ex0 = 0, st_ex1 = 0;
    auto stage_scalars = [&]() {
        st_second = st_c >= p.C0;
        const int ld2 = (st_second ? p.lda1 : p.lda0) * 2;
        st_add = (st_second ? st_c - p.C0 : st_c) * 2;
        if (CONV) {
            const int dy = st_tap / 3, dx = st_tap - dy * 3;
            st_bit = 1u << st_tap;
            if (UPS) { st_ld2 = ld2; st_ey0 = -(dy == 0); st_ey1 = (dy == 2); st_ex0 = -(dx == 0); st_ex1 = (dx == 2); }
            else st_add += ((dy - 1) * p.Ws + (dx - 1)) * ld2;
        }
    };
    stage_scalars();
    auto stage_advance = [&]() {
        st_k0 += 64;
        if (CONV) { if (++st_tap == 9) { st_tap = 0; st_c += 64; } }
        else st_c += 64;
        stage_scalars();
    };
    auto issue = [&](int j, int buf) {
        char* dst = smem + buf * STAGE + (j < 4 ? j * 8192 : A_BYTES + (j - 4) * 8192) + wid * 1024;
        if (j < 4) {
            unsigned voff = (st_second ? a_off1[j] : a_off0[j]) + (unsigned)st_add;
            if (CONV && UPS) {
                const int ey = (a_mask[j] >> 9) & 1 ? st_ey1 : st_ey0;
                const int ex = (a_mask[j] >> 10) & 1 ? st_ex1 : st_ex0;
                voff += (unsigned)(__mul24(__mul24(ey, p.Ws) + ex, st_ld2));
            }
            voff = (a_mask[j] & st_bit) ? voff : OOB;
            if (st_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, LDS_PTR(dst), 16, voff, 0, 0, 0);
            else           __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, LDS_PTR(dst), 16, voff, 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(dst), 16, w_off, (j - 4) * w_part + st_k0 * 2, 0, 0);
        }
    };
    // all parts of one K-tile (prologue)
    auto issue_all = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 6; ++j) issue(j, buf);
        if (grp == 0) issue(6, buf);
    };

    // ---- fragment read addresses (swizzled): A rows wr*64 + mt*16 + fr, W rows wc*80 + nt*16 + fr
    const int swz0 = ((fg) ^ (fr & 7)) << 4, swz1 = ((4 + fg) ^ (fr & 7)) << 4;
    const int a_rd0 = (wr * 64 + fr) * 128 + swz0, a_rd1 = (wr * 64 + fr) * 128 + swz1;
    const int b_rd0 = A_BYTES + (wc * WN + fr) * 128 + swz0, b_rd1 = A_BYTES + (wc * WN + fr) * 128 + swz1;

    floatx4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 bf[NT], af[MT];

    // waves 0-3 have 7 loads per K-tile in flight, waves 4-7 have 6: the counted wait leaves exactly the newest tile's
#define WAIT_PROLOGUE() do { if (grp == 0) wait_vmcnt<7>(); else wait_vmcnt<6>(); } while (0)

    // ---- prologue: K-tiles 0 and 1 into buffers 0 and 1
    issue_all(0);
    stage_advance();
    if (nk > 1) { issue_all(1); stage_advance(); WAIT_PROLOGUE(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind

#define LOAD_FRAGS(bo, ao)                                                                            \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const half8*)(cur + (bo) + nt * 2048);  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) af[mt] = *(const half8*)(cur + (ao) + mt * 2048);
#define ISSUE_L(sec)                                                                                  \
    if (more2) {                                                                                      \
        _Pragma("unroll") for (int i = 0; i < kMidSched[VAR][0][sec]; ++i) if (grp == 0) issue(sched_base(VAR, 0, sec) + i, nb_); \
        _Pragma("unroll") for (int i = 0; i < kMidSched[VAR][1][sec]; ++i) if (grp == 1) issue(sched_base(VAR, 1, sec) + i, nb_); \
    }
#define MFMA_BLOCK(sec)                                                                               \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int i = 0; i < NT * MT; ++i) {                                              \
        acc[i / MT][i % MT] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i / MT], af[i % MT], acc[i / MT][i % MT], 0, 0, 0); \
        _Pragma("unroll") for (int k = 0; k < kMidSched[VAR][0][sec]; ++k)                             \
            if (i == sched_slot(k, kMidSched[VAR][0][sec])) { FENCE(); if (more2 && grp == 0) issue(sched_base(VAR, 0, sec) + k, nb_); FENCE(); } \
        _Pragma("unroll") for (int k = 0; k < kMidSched[VAR][1][sec]; ++k)                             \
            if (i == sched_slot(k, kMidSched[VAR][1][sec])) { FENCE(); if (more2 && grp == 1) issue(sched_base(VAR, 1, sec) + k, nb_); FENCE(); } \
    }                                                                                                 \
    __builtin_amdgcn_s_setprio(0);
#define FENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef SDE_DIAG
    // in-kernel stamps (diagnostic build only).  -DSDE_DIAG=1: cycles per wave spent in each section of the two phases, summed
    // over K-tiles.  -DSDE_DIAG=2: only one pair of stamps around the loop (shader clock = d(memtime) / d(memrealtime) * 100 MHz).
    unsigned long long st_prev = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#if SDE_DIAG == 1
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[i] += (unsigned)(t_ - st_prev); st_prev = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#else
#define STAMP(i) do { } while (0)
#endif
    int cb = 0;   // ring slot of the tile being computed
    for (int kt = 0; kt < nk; ++kt) {
        const char* cur = smem + cb * STAGE;
        const int nb_ = cb == 0 ? 2 : cb - 1;          // slot of tile kt+2 == slot of tile kt-1
        const bool more2 = kt + 2 < nk;
        // phase 0 (k-half 0).  Load section L0: waves 0-3 may only overwrite rows nobody else reads (A rows 0-127).
        LOAD_FRAGS(b_rd0, a_rd0)
        ISSUE_L(0)
        FENCE(); STAMP(0); __builtin_amdgcn_s_barrier(); STAMP(1); FENCE();
        MFMA_BLOCK(1)
        FENCE(); STAMP(2); __builtin_amdgcn_s_barrier(); STAMP(3); FENCE();
        // phase 1 (k-half 1)
        LOAD_FRAGS(b_rd1, a_rd1)
        ISSUE_L(2)
        FENCE();
        // counted waits leave exactly what the wave has issued of tile kt+2 so far in flight: tile kt+1 has landed
        if (grp == 1) { if (more2) wait_vmcnt<sched_base(VAR, 1, 3)>(); else wait_vmcnt<0>(); }
        STAMP(4); __builtin_amdgcn_s_barrier(); STAMP(5); FENCE();
        MFMA_BLOCK(3)
        if (more2) stage_advance();   // scalar work of the next tile's addresses, under the tail of the MFMAs
        FENCE();
        if (grp == 0) { if (more2) wait_vmcnt<7>(); else wait_vmcnt<0>(); }
        STAMP(6); __builtin_amdgcn_s_barrier(); STAMP(7); FENCE();
        cb = cb == 2 ? 0 : cb + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the extra barrier of waves 4-7
#ifdef SDE_DIAG
    if (p.stamps && blockIdx.x == 8 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) p.stamps[wid * 8 + i] = st_sum[i];
        if (wid == 0) {
            p.stamps[64] = nk;
            p.stamps[65] = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0);
            p.stamps[66] = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
        }
    }
#endif
#undef STAMP
#undef LOAD_FRAGS
#undef MFMA_BLOCK
#undef FENCE
#undef WAIT_PROLOGUE
#undef ISSUE_L

    mid_epilogue<LO>(p, smem, acc, m0, n0, ks, z, wid, wr, wc, lane, fr, fg);
}

// ------------------------------------------------------------------------------------------ loader-wave variant
// 12 waves: waves 0-7 compute exactly as above but never touch global memory; waves 8-11 (one per SIMD, the kernel fits three
// waves per SIMD at <= 168 VGPRs) do nothing but issue the LDS-DMA.  A `buffer_load ... lds` costs its issuing wave ~100-150
// cycles; in the 8-wave kernel that cost sits in the compute waves' load sections and makes them longer than the MFMA sections
// they are paired with (section stamps).  Here the compute waves' load sections are nine ds_read_b128 only.
// All 12 waves execute the same barrier sequence; the loaders follow the timeline of waves 0-3.  Per K-tile (4 intervals):
//   I0: A rows 0-127 of tile t+2 (rows only waves 0-3 read: their reads of tile t-1 are long done)     4 pieces per loader
//   I1: A rows 128-255 (waves 4-7 finished reading tile t-1 during I0)                                  4
//   I2: W rows, first 3 pieces                                                                          3
//   I3: W rows, last 2 pieces; then wait until only tile t+2's 13 pieces are in flight (tile t+1 has landed)
//
// BM_ = 128 (linears of the 16x16 level, M = 4096: 256 tiles of 128x160 where 256-row tiles would fill half the chip and the 4-wave
// 64x160 kernel moves 1.6x the bytes through the L2 -> LDS path, which is what bounds it): 4 compute waves (2 x 2, the same 64x80 wave
// tile; waves 0-1 = group 0, waves 2-3 one barrier behind) + 4 loaders, and the smaller stage buys a 4-deep ring: loads run THREE
// tiles ahead.  Same interval plan: I0 A rows 0-63 (read by group 0 only), I1 A rows 64-127, I2 / I3 the W rows.
template <bool CONV, bool UPS, bool LO = false, int BM_ = 256>
__global__ __launch_bounds__((BM_ / 32 + 4) * 64) void gemm_midl_kernel(GemmDev p)
{
    constexpr int NCW = BM_ / 32;                 // compute waves
    constexpr int NQ = BM_ / 32;                  // A pieces (8 rows x 128 B) per loader and K-tile
    constexpr int A_B = BM_ * 128, STG = A_B + BN * 128;
    constexpr int NSTG = BM_ == 256 ? 3 : 4, DIST = NSTG - 1;   // ring depth; loads run DIST tiles ahead
    constexpr int PP = NQ + 5;                    // pieces per loader and K-tile
    static_assert(NSTG * STG <= 160 * 1024, "LDS");
    constexpr int BM = BM_;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int z = blockIdx.z;

    const int nwg = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = p.colmajor ? bid % p.nbm : bid / p.nbn, tn = p.colmajor ? bid / p.nbm : bid - (bid / p.nbn) * p.nbn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk_all = p.K >> 6;
    const int ks = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * ks / p.splitk);
    const int nk = (int)((long long)nk_all * (ks + 1) / p.splitk) - kt0;
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { FENCE(); __builtin_amdgcn_s_barrier(); FENCE(); } while (0)

    if (wid >= NCW) {
        // ================================================================ loader waves
        const int lw = wid - NCW;
        const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A0 + z * p.sA), 0, OOB, RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)((p.A1 ? p.A1 : p.A0) + z * p.sA), 0, OOB, RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + z * p.sW), 0, OOB, RSRC_FLAGS);
        const int srow = lane >> 3;
        const unsigned lchunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
        // A piece q = 2*j + i (j = 64-row part, i = 0/1): rows j*64 + (lw*2 + i)*8 + srow
        unsigned a_off0[NQ], a_off1[NQ], a_mask[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int m = m0 + (q >> 1) * 64 + (lw * 2 + (q & 1)) * 8 + srow;
            const bool ok = m < p.M;
            int pix = m;
            if (CONV) {
                const int hw = p.Ho * p.Wo;
                const int nb = m / hw, rem = m - nb * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int cy = oy * p.stride + p.coff, cx = ox * p.stride + p.coff;
                unsigned mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = cy + t / 3 - 1, ix = cx + t % 3 - 1;
                    if (ok && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) mask |= 1u << t;
                }
                mask |= (unsigned)(cy & 1) << 9 | (unsigned)(cx & 1) << 10;
                a_mask[q] = mask;
                pix = nb * p.Hs * p.Ws + (cy >> (UPS ? 1 : 0)) * p.Ws + (cx >> (UPS ? 1 : 0));
            } else {
                a_mask[q] = ok ? 1u : 0u;
            }
            a_off0[q] = (unsigned)pix * (unsigned)(p.lda0 * 2) + lchunk16;
            a_off1[q] = (unsigned)pix * (unsigned)(p.lda1 * 2) + lchunk16;
        }
        // W piece i (0..4): rows (lw*5 + i)*8 + srow
        const unsigned w_off = (unsigned)((n0 + lw * 40 + srow) * p.K) * 2u + lchunk16;
        const int w_step = 8 * p.K * 2;

        int st_k0 = kt0 << 6, st_tap = 0, st_c = st_k0;
        if (CONV) { st_tap = kt0 % 9; st_c = (kt0 / 9) << 6; }
        bool st_second = false;
        unsigned st_bit = 1u;
        int st_add = 0, st_ld2 = 0, st_ey0 = 0, st_ey1 = 0, st_ex0 = 0, st_ex1 = 0;
        auto stage_scalars = [&]() {
            st_second = st_c >= p.C0;
            const int ld2 = (st_second ? p.lda1 : p.lda0) * 2;
            st_add = (st_second ? st_c - p.C0 : st_c) * 2;
            if (CONV) {
                const int dy = st_tap / 3, dx = st_tap - dy * 3;
                st_bit = 1u << st_tap;
                if (UPS) { st_ld2 = ld2; st_ey0 = -(dy == 0); st_ey1 = (dy == 2); st_ex0 = -(dx == 0); st_ex1 = (dx == 2); }
                else st_add += ((dy - 1) * p.Ws + (dx - 1)) * ld2;
            }
        };
        stage_scalars();
        auto stage_advance = [&]() {
            st_k0 += 64;
            if (CONV) { if (++st_tap == 9) { st_tap = 0; st_c += 64; } }
            else st_c += 64;
            stage_scalars();
        };
        auto issue_a = [&](int q, int buf) {
            char* dst = smem + buf * STG + (q >> 1) * 8192 + (lw * 2 + (q & 1)) * 1024;
            unsigned voff = (st_second ? a_off1[q] : a_off0[q]) + (unsigned)st_add;
            if (CONV && UPS) {
                const int ey = (a_mask[q] >> 9) & 1 ? st_ey1 : st_ey0;
                const int ex = (a_mask[q] >> 10) & 1 ? st_ex1 : st_ex0;
                voff += (unsigned)(__mul24(__mul24(ey, p.Ws) + ex, st_ld2));
            }
            voff = (a_mask[q] & st_bit) ? voff : OOB;
            if (st_second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, LDS_PTR(dst), 16, voff, 0, 0, 0);
            else           __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, LDS_PTR(dst), 16, voff, 0, 0, 0);
        };
        auto issue_w = [&](int i, int buf) {
            char* dst = smem + buf * STG + A_B + (lw * 5 + i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(dst), 16, w_off, i * w_step + st_k0 * 2, 0, 0);
        };
        auto issue_all = [&](int buf) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) issue_a(q, buf);
#pragma unroll
            for (int i = 0; i < 5; ++i) issue_w(i, buf);
        };
        if constexpr (BM_ == 128) {
            // one barrier per K-tile (see the compute waves below): barrier kt says "tile kt+1 has landed" to the compute waves and
            // "nobody reads tile kt any more" to the loaders, who refill its slot with tile kt+4 straight away.
#pragma unroll
            for (int s = 0; s < NSTG; ++s) if (s < nk) { issue_all(s); stage_advance(); }
            if (nk >= 4) wait_vmcnt<3 * PP>(); else if (nk == 3) wait_vmcnt<2 * PP>(); else if (nk == 2) wait_vmcnt<PP>(); else wait_vmcnt<0>();
            BAR();
            int cb = 0;
            for (int kt = 0; kt < nk; ++kt) {
                const int ahead = min(nk - 1, kt + 3) - (kt + 1);   // tiles after kt+1 that are in flight
                if (ahead >= 2) wait_vmcnt<2 * PP>(); else if (ahead == 1) wait_vmcnt<PP>(); else wait_vmcnt<0>();
                BAR();
                if (kt + NSTG < nk) { issue_all(cb); stage_advance(); }
                cb = cb == NSTG - 1 ? 0 : cb + 1;
            }
            return;
        }
        // prologue: the first DIST tiles; wait until tile 0 has landed (everything but the newer tiles' pieces)
        issue_all(0);
        stage_advance();
        if (nk > 1) { issue_all(1); stage_advance(); }
        if (DIST == 3 && nk > 2) { issue_all(2); stage_advance(); }
        if (DIST == 3 && nk > 2) wait_vmcnt<2 * PP>();
        else if (nk > 1) wait_vmcnt<PP>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int cb = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int nb_ = cb == 0 ? NSTG - 1 : cb - 1;   // slot of tile kt+DIST == slot of tile kt-1
            const bool more2 = kt + DIST < nk;
            if (more2) {
#pragma unroll
                for (int q = 0; q < NQ / 2; ++q) issue_a(q, nb_);
            }
            BAR();
            if (more2) {
#pragma unroll
                for (int q = NQ / 2; q < NQ; ++q) issue_a(q, nb_);
            }
            BAR();
            if (more2) { issue_w(0, nb_); issue_w(1, nb_); issue_w(2, nb_); }
            BAR();
            // leave only the tiles after kt+1 in flight: tile kt+1 has landed when the interval ends
            if (more2) { issue_w(3, nb_); issue_w(4, nb_); stage_advance(); wait_vmcnt<(DIST - 1) * PP>(); }
            else if (DIST == 3 && kt + 2 < nk) wait_vmcnt<PP>();
            else wait_vmcnt<0>();
            BAR();
            cb = cb == NSTG - 1 ? 0 : cb + 1;
        }
        __builtin_amdgcn_s_barrier();   // the extra barrier of waves 4-7
        return;
    }

    // ==================================================================== compute waves
    const int wr = wid >> 1, wc = wid & 1, grp = wid >= NCW / 2;
    const int fr = lane & 15, fg = lane >> 4;
    const int swz0 = ((fg) ^ (fr & 7)) << 4, swz1 = ((4 + fg) ^ (fr & 7)) << 4;
    const int a_rd0 = (wr * 64 + fr) * 128 + swz0, a_rd1 = (wr * 64 + fr) * 128 + swz1;
    const int b_rd0 = A_B + (wc * WN + fr) * 128 + swz0, b_rd1 = A_B + (wc * WN + fr) * 128 + swz1;
    floatx4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 bf[NT], af[MT];
    if constexpr (BM_ == 128) {
        // One compute wave per SIMD: nothing to ping-pong with, so each wave pipelines by itself.  Two fragment sets; the LDS reads of
        // the next k-half are issued in front of the 20 MFMAs of the current one, and a K-tile costs one barrier instead of four.
        half8 bg[NT], ag[MT];
#define LOAD_SET(B_, A_, base, bo, ao)                                                                   \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) B_[nt] = *(const half8*)((base) + (bo) + nt * 2048); \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) A_[mt] = *(const half8*)((base) + (ao) + mt * 2048);
#define MFMA_SET(B_, A_)                                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                   \
        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                 \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                             \
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(B_[nt], A_[mt], acc[nt][mt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);
        BAR();                                   // tile 0 has landed
        LOAD_SET(bf, af, smem, b_rd0, a_rd0)
        int cb = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const char* cur = smem + cb * STG;
            cb = cb == NSTG - 1 ? 0 : cb + 1;
            LOAD_SET(bg, ag, cur, b_rd1, a_rd1)
            FENCE();
            MFMA_SET(bf, af)
            FENCE();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tile kt are complete
            BAR();
            if (kt + 1 < nk) { LOAD_SET(bf, af, smem + cb * STG, b_rd0, a_rd0) }
            FENCE();
            MFMA_SET(bg, ag)
            FENCE();
        }
#undef LOAD_SET
#undef MFMA_SET
        mid_epilogue<LO, NSTG * STG / NCW>(p, smem, acc, m0, n0, ks, z, wid, wr, wc, lane, fr, fg);
        return;
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind
#define LOAD_FRAGS(bo, ao)                                                                            \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const half8*)(cur + (bo) + nt * 2048);  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) af[mt] = *(const half8*)(cur + (ao) + mt * 2048);
#define MFMA_ALL()                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                              \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[mt], acc[nt][mt], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
    int cb = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* cur = smem + cb * STG;
        LOAD_FRAGS(b_rd0, a_rd0)
        BAR();
        MFMA_ALL()
        BAR();
        LOAD_FRAGS(b_rd1, a_rd1)
        BAR();
        MFMA_ALL()
        BAR();
        cb = cb == NSTG - 1 ? 0 : cb + 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the extra barrier of waves 4-7
#undef LOAD_FRAGS
#undef MFMA_ALL
#undef BAR
#undef FENCE
    mid_epilogue<LO, NSTG * STG / NCW>(p, smem, acc, m0, n0, ks, z, wid, wr, wc, lane, fr, fg);
}

// ------------------------------------------------------------------------------------------ halo variant (3x3, stride 1)
// The implicit GEMM above fetches every input pixel nine times, once per tap, and the L2 -> LDS path (~70 GB/s per CU) is what the
// 256x160 convolutions run against: 53 KB per K-tile = 0.76 us where the 40 MFMAs per wave need 0.53.  Here a tile is a 16 x 16 PATCH
// of output pixels of one image, and the 18 x 18 input halo of one 64-channel chunk is staged ONCE (324 rows of 128 B) and read by
// all nine taps at row offset dy * 18 + dx: 4.6 KB of A per K-tile instead of 32 KB, 25 KB per K-tile in all.
//   * K order is the other kernels' (chunk, tap, 64 channels): same weight buffer.
//   * LDS: 2 halo buffers x 352 rows (44 pieces of 8 rows; rows >= 324 and out-of-image rows are zero-filled by the range check)
//     + 3 W stages x 160 rows = 151,552 B.  Same XOR swizzle, keyed on the halo row.
//   * 8 compute waves exactly as gemm_midl_kernel (64 x 80 wave tiles, four barriers per K-tile, waves 4-7 one barrier behind): a
//     wave's 64 rows are four patch rows of 16 pixels.  4 loader waves, at most two pieces per interval each: the NEXT chunk's halo
//     (11 pieces per loader) in interval 0 of taps 1-7, W of tile t+2 (5 pieces) in intervals 1-3.  The counted wait at the end of a
//     K-tile leaves only that tile's own issues in flight, so W of tile t+1 has landed and, by the end of tap 8, the whole halo.
//   * Split-K splits at chunk boundaries.
namespace {
constexpr int H_ROWS = 352, H_BYTES = H_ROWS * 128;   // 45,056 B per halo buffer
constexpr int H_PIECES = 11;                          // halo pieces per loader wave
constexpr int W_STG = BN * 128;                       // 20,480 B
constexpr int H_SMEM = 2 * H_BYTES + 3 * W_STG;       // 151,552 B
// pieces of the next chunk's halo issued at each tap, in interval 0: none at tap 0 (waves 4-7 are still reading that buffer) and none at
// tap 8 (a tile's own issues are only known to have landed one tile later)
constexpr int halo_cnt(int tap) { return tap == 0 || tap == 8 ? 0 : (tap <= 4 ? 2 : 1); }   // 0,2,2,2,2,1,1,1,0
constexpr int halo_first(int tap) { return tap <= 4 ? 2 * (tap - 1) : 3 + tap; }            // -,0,2,4,6,8,9,10,-
// UPS (fused nearest-2x upsample): the patch's halo in SOURCE pixels is 10 x 10 = 100 rows = 13 pieces, four per loader (taps 1-2)
constexpr int halo_cnt_ups(int tap) { return tap == 1 || tap == 2 ? 2 : 0; }
constexpr int halo_first_ups(int tap) { return 2 * (tap - 1); }
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }
}  // namespace

// UPS: the input is the nearest-2x upsample of a (Hi/2 x Wi/2) source.  Virtual input pixel (y, x) is source pixel (y >> 1, x >> 1), so the
// 16 x 16 patch at (Y0, X0) needs source rows Y0/2 - 1 .. Y0/2 + 8: a 10 x 10 halo, and pixel (py, px) reads tap (dy, dx) at halo row
// ((py + dy + 1) >> 1) * 10 + ((px + dx + 1) >> 1).  Virtual rows -1 and Hi fall on source rows -1 and Hi/2: the range-checked zero fill.
// NT_ = 4: 256 x 128 block tile (wave tile 64 x 64) for widths that are multiples of 128 but not of 160 (the VAE decoder's 128-channel
// convolutions at 512 x 512 ran on the 4-wave 128x128 kernel at 0.5 PFLOP/s: 32 KB through L2 -> LDS per 2.1 MFLOP K-tile).
#ifndef MIDH_RG
#define MIDH_RG 1
#endif
template <bool LO, bool UPS = false, int NT_ = 5>
__global__ __launch_bounds__(768) void gemm_midh_kernel(GemmDev p)
{
    constexpr int NT = NT_, WN = NT_ * 16, BN = 2 * WN, W_STG = BN * 128;   // shadow the file-scope constants
    constexpr int NWP = BN / 32;                 // W pieces per loader and K-tile (5 or 4)
    constexpr int NP = UPS ? 4 : H_PIECES;       // halo pieces per loader
    constexpr int HWD = UPS ? 10 : 18;           // halo width
    constexpr int HN = HWD * HWD;                // halo rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = p.colmajor ? bid % p.nbm : bid / p.nbn, tn = p.colmajor ? bid / p.nbm : bid - (bid / p.nbn) * p.nbn;
    const int npx = p.Wo >> 4, ppi = (p.Ho >> 4) * npx;         // patches per row / per image
    const int img = tm / ppi, pr = tm - img * ppi;
    const int Y0 = (pr / npx) << 4, X0 = (pr - (pr / npx) * npx) << 4;
    const int n0 = tn * BN;
    const int nch = p.K / 576;                                   // 64-channel chunks
    const int ks = blockIdx.y;
    const int c0 = (int)((long long)nch * ks / p.splitk), c1 = (int)((long long)nch * (ks + 1) / p.splitk);
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { FENCE(); __builtin_amdgcn_s_barrier(); FENCE(); } while (0)
    char* const wbase = smem + 2 * H_BYTES;

    if (wid >= 8) {
        // ================================================================ loader waves
        const int lw = wid - 8;
        const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A0, 0, OOB, RSRC_FLAGS);
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, OOB, RSRC_FLAGS);
        const int srow = lane >> 3;
        const unsigned lchunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
        // halo piece j of this loader: rows (lw + 4j) * 8 + srow of the 18 x 18 halo (row h = hy * 18 + hx)
        unsigned hpix[NP], hmask = 0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int h = (lw + 4 * j) * 8 + srow;
            const int hy = h / HWD, hx = h - hy * HWD;
            const int iy = (UPS ? (Y0 >> 1) : Y0) - 1 + hy, ix = (UPS ? (X0 >> 1) : X0) - 1 + hx;   // source pixel (p.Hs x p.Ws; == Hi x Wi without UPS)
            const bool ok = h < HN && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
            hpix[j] = ok ? (unsigned)((img * p.Hs + iy) * p.Ws + ix) : 0u;
            hmask |= (ok ? 1u : 0u) << j;
        }
        // one source tensor only: no 3x3 convolution of the models reads a channel concat through two pointers (the GroupNorm in
        // front of it has already written the concatenated, normalised tensor); gemm_midh_supported refuses A1
        const unsigned ld2 = (unsigned)(p.lda0 * 2);
        auto issue_halo = [&](int j, int c, int hb) {
            unsigned voff = hpix[j] * ld2 + lchunk16 + (unsigned)(c << 7);
            voff = (hmask >> j) & 1u ? voff : OOB;
            char* dst = smem + hb * H_BYTES + (lw + 4 * j) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, LDS_PTR(dst), 16, voff, 0, 0, 0);
        };
        const unsigned w_off = (unsigned)((n0 + lw * (NWP * 8) + srow) * p.K) * 2u + lchunk16;
        const int w_step = 8 * p.K * 2;
        int st_k0 = c0 * 576;
        auto issue_w = [&](int i, int slot) {
            char* dst = wbase + slot * W_STG + (lw * NWP + i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(dst), 16, w_off, i * w_step + st_k0 * 2, 0, 0);
        };
        // prologue: the first chunk's halo, W of tiles 0 and 1 (a chunk is nine tiles: both exist)
#pragma unroll
        for (int j = 0; j < NP; ++j) issue_halo(j, c0, 0);
#pragma unroll
        for (int i = 0; i < NWP; ++i) issue_w(i, 0);
        st_k0 += 64;
#pragma unroll
        for (int i = 0; i < NWP; ++i) issue_w(i, 1);
        st_k0 += 64;
        wait_vmcnt<NWP>();
        __builtin_amdgcn_s_barrier();
        int hb = 0;
        for (int c = c0; c < c1; ++c) {
            const bool next = c + 1 < c1;
            static_for<9>([&](auto tap_) {
                constexpr int tap = decltype(tap_)::value;
                const bool more2 = next || tap < 7;              // tile + 2 exists
                const int slot = (tap + 2) % 3;                  // == slot of tile - 1
                // at most two pieces per interval: an LDS-DMA costs its issuing wave 100-150 cycles, and an interval is 20 MFMAs (320)
                constexpr int hcnt = UPS ? halo_cnt_ups(tap) : halo_cnt(tap), hfirst = UPS ? halo_first_ups(tap) : halo_first(tap);
                if (next) {
#pragma unroll
                    for (int q = 0; q < hcnt; ++q) issue_halo(hfirst + q, c + 1, hb ^ 1);
                }
                BAR();
                if (more2) { issue_w(0, slot); issue_w(1, slot); }
                BAR();
                if (more2) { issue_w(2, slot); issue_w(3, slot); }
                BAR();
                if (more2) { if (NWP == 5) issue_w(4, slot); st_k0 += 64; }
                // in flight afterwards: this tile's own issues only (the halo pieces above, W of tile + 2)
                if (more2) { if (next) wait_vmcnt<NWP + hcnt>(); else wait_vmcnt<NWP>(); }
                else wait_vmcnt<0>();
                BAR();
            });
            hb ^= 1;
        }
        __builtin_amdgcn_s_barrier();   // the extra barrier of waves 4-7
        return;
    }

    // ==================================================================== compute waves
    const int wr = wid >> 1, wc = wid & 1, grp = wid >> 2;
    const int fr = lane & 15, fg = lane >> 4;
    const int swz0 = ((fg) ^ (fr & 7)) << 4, swz1 = ((4 + fg) ^ (fr & 7)) << 4;
    const int b_rd0 = (wc * WN + fr) * 128 + swz0, b_rd1 = (wc * WN + fr) * 128 + swz1;
    // halo row of this lane's pixel in patch row wr*4, tap (0,0); UPS: the lane's pixel column alone (rows and columns are halved per tap)
    const int hrow0 = UPS ? fr : wr * 4 * 18 + fr;
    floatx4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 bf[NT], af[MT];
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind
#define LOAD_FRAGS_H(bo, kc, tapoff)                                                                  \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const half8*)(curw + (bo) + nt * 2048);  \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                \
        const int h_ = UPS ? ((wr * 4 + mt + tap / 3 + 1) >> 1) * 10 + ((hr + tap % 3 + 1) >> 1) : hr + mt * 18 + (tapoff);   \
        af[mt] = *(const half8*)(curh + (h_ << 7) + ((((kc) ^ h_) & 7) << 4));                         \
    }
#define MFMA_ALL()                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                              \
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[mt], acc[nt][mt], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
    int hb = 0;
    for (int c = c0; c < c1; ++c) {
        const char* curh = smem + hb * H_BYTES;
        // the 72 fragment addresses of a chunk do not depend on the chunk: left alone the compiler hoists them all out of the loop and
        // spills them.  Three VALU instructions per read are cheaper than a scratch reload.
        int hr = hrow0;
        asm volatile("" : "+v"(hr));
        static_for<9>([&](auto tap_) {
            constexpr int tap = decltype(tap_)::value;
            const char* curw = wbase + (tap % 3) * W_STG;
            constexpr int tapoff = (tap / 3) * 18 + tap % 3;
            LOAD_FRAGS_H(b_rd0, fg, tapoff)
            BAR();
            MFMA_ALL()
            BAR();
            LOAD_FRAGS_H(b_rd1, 4 + fg, tapoff)
            BAR();
            MFMA_ALL()
            BAR();
        });
        hb ^= 1;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the extra barrier of waves 4-7
#undef LOAD_FRAGS_H
#undef MFMA_ALL
#undef BAR
#undef FENCE
    const int pbase = (img * p.Ho + Y0) * p.Wo + X0;
    // 16-row epilogue steps on the 160-column tiles (32-row steps spill 35 registers here: 3 us per launch); the 128-column tiles have room
    mid_epilogue<LO, H_SMEM / 8, true, NT_, (NT_ == 5 ? MIDH_RG : 2)>(p, smem, acc, pbase, n0, ks, 0, wid, wr, wc, lane, fr, fg);
}

// ------------------------------------------------------------------------------------------ host side
bool gemm_big_supported(const GemmArgs& a);   // gemm_big.hip: shared operand rules

static bool mid_operands_ok(const GemmArgs& a, int bn)
{
    if (a.epi != EPI_STD || a.Npad % bn || (a.N & 7)) return false;
    if ((a.Rlo || a.Dlo) && ((a.ldd & 7) || (a.R && (a.ldr & 7)))) return false;   // the split-fp16 epilogue has 16-byte paths only
    GemmArgs c = a;
    c.Npad = 320 * ((a.Npad + 319) / 320);   // reuse the operand / offset-range rules, not the tile-width rule
    return gemm_big_supported(c);
}
bool gemm_mid_supported(const GemmArgs& a) { return mid_operands_ok(a, BN); }

template <bool CONV, bool UPS, int VAR = 0>
static void launch_mid_t(const LaunchCtx& ctx, const GemmDev& p, dim3 grid)
{
    if (VAR == 0 && (p.Rlo || p.Dlo)) {
        auto kern = gemm_mid_kernel<CONV, UPS, 0, true>;
        allow_dynamic_lds((const void*)kern, SMEM_BYTES);
        hipLaunchKernelGGL(kern, grid, dim3(512), SMEM_BYTES, ctx.stream, p);
        return;
    }
    auto kern = gemm_mid_kernel<CONV, UPS, VAR>;
    allow_dynamic_lds((const void*)kern, SMEM_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(512), SMEM_BYTES, ctx.stream, p);
}

template <bool CONV, bool UPS>
static void launch_midl_t(const LaunchCtx& ctx, const GemmDev& p, dim3 grid)
{
    if (p.Rlo || p.Dlo) {
        auto kern = gemm_midl_kernel<CONV, UPS, true>;
        allow_dynamic_lds((const void*)kern, SMEM_BYTES);
        hipLaunchKernelGGL(kern, grid, dim3(768), SMEM_BYTES, ctx.stream, p);
        return;
    }
    auto kern = gemm_midl_kernel<CONV, UPS>;
    allow_dynamic_lds((const void*)kern, SMEM_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(768), SMEM_BYTES, ctx.stream, p);
}

// 3x3 stride-1 convolutions on whole 16 x 16 output patches: the halo kernel
bool gemm_midh_supported(const GemmArgs& a)
{
    if (a.taps != 9 || a.stride != 1 || a.coff || a.batch != 1 || a.A1 || a.C1) return false;
    static const int ups_env = getenv("SDE_HALOUPS") ? atoi(getenv("SDE_HALOUPS")) : 1;   // tuning hook: 0 = upsampling convs stay on the per-tap kernels
    if (a.ups && !ups_env) return false;
    // a.ups: Hi x Wi is the virtual (upsampled) size; the cropped variant (2H - 1) is odd and fails the patch rule
    if (a.Ho != a.Hi || a.Wo != a.Wi || (a.Ho & 15) || (a.Wo & 15) || a.M % (a.Ho * a.Wo)) return false;
    if (a.K % 576) return false;
    return mid_operands_ok(a, a.Npad % BN == 0 ? BN : 128);   // 160-column tiles, else 128-column ones
}

void launch_gemm_midh(const LaunchCtx& ctx, GemmDev& p)
{
    const bool n160 = p.Npad % BN == 0;
    p.nbm = p.M / BM;
    p.nbn = p.Npad / (n160 ? BN : 128);
    SDE_REQUIRE(p.splitk >= 1 && p.splitk <= p.K / 576, SDE_ERR_INVALID_ARGUMENT, "launch_gemm_midh: more split-K ways than 64-channel chunks");
    dim3 grid(p.nbm * p.nbn, p.splitk, 1);
    auto go = [&](auto kern) {
        allow_dynamic_lds((const void*)kern, H_SMEM);
        hipLaunchKernelGGL(kern, grid, dim3(768), H_SMEM, ctx.stream, p);
    };
    const bool lo = p.Rlo || p.Dlo;
    if (!n160) {   // VAE widths: no split-fp16 tensors there
        SDE_REQUIRE(!lo, SDE_ERR_INVALID_ARGUMENT, "launch_gemm_midh: split-fp16 epilogue needs a 160-column tiling");
        if (p.ups) go(gemm_midh_kernel<false, true, 4>); else go(gemm_midh_kernel<false, false, 4>);
        return;
    }
    if (p.ups) { if (lo) go(gemm_midh_kernel<true, true>); else go(gemm_midh_kernel<false, true>); }
    else       { if (lo) go(gemm_midh_kernel<true, false>); else go(gemm_midh_kernel<false, false>); }
}

// 128x160 tiles (linears only): 4 compute + 4 loader waves, 4-deep ring
void launch_gemm_mid128(const LaunchCtx& ctx, GemmDev& p, int batch)
{
    constexpr int BM128 = 128, SMEM128 = 4 * (BM128 * 128 + BN * 128);   // 147,456 B
    p.nbm = cdiv(p.M, BM128);
    p.nbn = p.Npad / BN;
    p.splitk = 1; p.ws = nullptr;
    dim3 grid(p.nbm * p.nbn, 1, batch);
    if (p.Rlo || p.Dlo) {
        auto kern = gemm_midl_kernel<false, false, true, BM128>;
        allow_dynamic_lds((const void*)kern, SMEM128);
        hipLaunchKernelGGL(kern, grid, dim3(512), SMEM128, ctx.stream, p);
    } else {
        auto kern = gemm_midl_kernel<false, false, false, BM128>;
        allow_dynamic_lds((const void*)kern, SMEM128);
        hipLaunchKernelGGL(kern, grid, dim3(512), SMEM128, ctx.stream, p);
    }
}

void launch_gemm_mid(const LaunchCtx& ctx, GemmDev& p, int batch, bool conv)
{
    p.nbm = cdiv(p.M, BM);
    p.nbn = p.Npad / BN;
    dim3 grid(p.nbm * p.nbn, p.splitk, batch);
    // loader waves: +14-17 % on the 3x3 convolutions (long K); the short-K linears lose 2-5 % to the heavier prologue
    static const int ldw = getenv("SDE_MIDLDW") ? atoi(getenv("SDE_MIDLDW")) : 1;   // tuning hook: 0 never, 1 convolutions, 2 always
    if (ldw == 2 || (ldw == 1 && conv)) {
        if (conv && p.ups) launch_midl_t<true, true>(ctx, p, grid);
        else if (conv)     launch_midl_t<true, false>(ctx, p, grid);
        else               launch_midl_t<false, false>(ctx, p, grid);
        return;
    }
#ifdef SDE_DIAG
    // schedule experiments (diagnostic builds): SDE_MIDVAR=1..3 selects kMidSched[v] for plain 3x3 convs
    static const int var = getenv("SDE_MIDVAR") ? atoi(getenv("SDE_MIDVAR")) : 0;
    if (conv && !p.ups && var == 1) return launch_mid_t<true, false, 1>(ctx, p, grid);
    if (conv && !p.ups && var == 2) return launch_mid_t<true, false, 2>(ctx, p, grid);
    if (conv && !p.ups && var == 3) return launch_mid_t<true, false, 3>(ctx, p, grid);
#endif
    if (conv && p.ups) launch_mid_t<true, true>(ctx, p, grid);
    else if (conv)     launch_mid_t<true, false>(ctx, p, grid);
    else               launch_mid_t<false, false>(ctx, p, grid);
}
