// gemm_ws.hip — W-stationary, persistent fp16 GEMM for the short-K linears (K <= 640) of the 64x64 / 32x32 UNet levels, gfx950.
//
// Same contract as the other GEMM kernels (D = epilogue(A * W^T), EPI_STD with optional residual / split-fp16 pairs, or the
// pairwise GEGLU).  These launches (1x1 / Linear projections over 16k-65k tokens with K = 320 or 640: proj_in, q / k / v, to_out,
// the GEGLU projection, proj_out, the ResNet shortcuts) move far more bytes than they multiply: M = 65536, N = K = 320 reads 42 MB
// and writes 42 MB for 13 GFLOP.  On the tile kernels (one 256x320 tile per CU) every CU runs its load, MFMA and store phases in
// lockstep with all the others, so HBM sees a read burst, an idle gap and a write burst: 36 us where a streaming pass over the
// same bytes takes 21.  Structure here:
//   * one block per CU, 8 waves, ALL INDEPENDENT in the steady state: no block barrier inside the row loop.
//   * The block keeps ONE W panel (PN columns x K, <= 100 KB: 160 columns at K <= 320, 80 at K = 640) resident in LDS, loaded
//     once per pass by LDS-DMA in the other kernels' swizzled image, and streams rows past it.
//   * A never touches LDS: each wave owns 32-row tiles and loads its A fragments straight into registers in MFMA operand
//     layout (16 B per lane, 64 contiguous bytes per row and instruction).  The registers are a ring of five K-tiles: as soon as
//     the MFMAs that read a slot have issued, the slot is refilled with the K-tile five further down the wave's stream (the next
//     tile's, at K <= 320), so a wave has 20 KB in flight while it computes and stores: 160 KB per CU, enough to cover HBM
//     latency at the chip's rate.
//   * Epilogue through a private 5.5 KB LDS slice per wave (fp32, 16 rows x 80 columns per step), 16-byte row-contiguous stores.
//     With eight waves at different points of their tiles a CU reads, multiplies and stores at the same time.
//   * Work split: 256-row chunks (8 waves x 32 rows).  Each XCD owns a contiguous range of chunks; inside an XCD `npp` adjacent
//     blocks hold different panels and walk the same chunks at the same time, so a chunk of A leaves HBM once and the other
//     panels' blocks hit the XCD's L2.  N / PN > npp: several passes, W reloaded per pass.
#include "gemm_common.h"

#include <algorithm>
#include <type_traits>
#include <vector>

namespace {
constexpr unsigned OOB = 0x80000000u;
constexpr int RSRC_FLAGS = 0x00020000;
constexpr int WS_WAVES = 8;                                       // 16 in the <.., WS_EPI_H16, false, 16> instances
constexpr int WS_STG_PITCH = 80 * 4 + 32;                         // fp32 staging row (80 columns) + pad
constexpr int WS_STG_BYTES = 16 * WS_STG_PITCH;                   // per wave: 5,632 B
constexpr int ws_panel_bytes(int ktmax, int nt) { return ktmax * nt * 16 * 128; }
constexpr int ws_bias_bytes(int nt) { return nt * 16 * 4; }
constexpr int WS_STG16_PITCH = 80 * 2 + 16;                       // fp16 staging row (80 columns) + pad
constexpr int WS_STG16_BYTES = 16 * WS_STG16_PITCH;               // per wave: 2,816 B
constexpr int ws_smem_bytes(int ktmax, int nt, int waves = WS_WAVES, int stg = WS_STG_BYTES) { return ws_panel_bytes(ktmax, nt) + ws_bias_bytes(nt) + waves * stg; }
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
enum { WS_EPI_STD = 0, WS_EPI_LO = 1, WS_EPI_GEGLU = 2, WS_EPI_H16 = 3 };   // H16: EPI_STD without a residual, staged as fp16
}  // namespace

// KT: K-tiles of 64 (5: K = 320).  NT: 16-column MFMA tiles per panel (panel width PN = 16 NT).  MT: 16-row tiles per wave tile.
// HAS_R: residual.  The A fragments of a tile (KT x 2 x MT registers quads) are refilled with the NEXT tile's as they are consumed.
// (K = 640 was built too -- 80-column panels, the ring turning twice per tile -- and measured 25-40 % SLOWER than the 256x160 /
// 256x320 tile kernels on every shape: with panels that narrow A is re-read N / 80 times out of L2; removed.)
// The tile body is straight-line code: hipcc's s_waitcnt insertion counts vmcnt exactly only without control flow between a load and
// its use -- with a branch around the prefetch (or around a store) every tile began with vmcnt(0), i.e. with a wait for its own
// stores.  So the last tile of a wave prefetches too (out-of-range offsets: the buffer range check answers with zeros, no memory
// access), tails are masked the same way on the store side, and the second A source is a scalar select, not a branch.
template <int KT, int NT, int MT, int EPI, bool HAS_R, int WAVES = WS_WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_ws_kernel(GemmDev p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PN = NT * 16, RD = 5;
    constexpr int STG = EPI == WS_EPI_H16 ? WS_STG16_BYTES : WS_STG_BYTES;
    constexpr int TROWS = 16 * MT, CHUNK = WAVES * TROWS;         // rows per wave tile / per block round
    static_assert(KT == 5, "K = 320");
    constexpr int PANEL = ws_panel_bytes(KT, NT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    // ---- work split (header)
    const int np = p.Npad / PN;                                   // panels
    const int npp = p.ws_npp;                                     // panels per pass (<= blocks per XCD)
    const int G = (int)gridDim.x >> 3;                            // blocks per XCD (blocks b and b + 8 share one)
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int slots = G / npp;
    const bool active = idx < slots * npp;
    const int slot = idx / npp, pin = idx - slot * npp;           // which chunks, which panel of the pass
    const int nchunks = (p.M + CHUNK - 1) / CHUNK;
    const int c_lo = (int)(((long long)nchunks * xcd) >> 3), c_hi = (int)(((long long)nchunks * (xcd + 1)) >> 3);
    const int npass = (np + npp - 1) / npp;

    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)p.D, 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rDl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Dlo ? p.Dlo : p.D), 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.R ? p.R : p.D), 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rRl = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Rlo ? p.Rlo : (p.R ? p.R : p.D)), 0, OOB, RSRC_FLAGS);
    const int kt_second = p.A1 ? (p.C0 >> 6) : KT + 1;             // first K-tile that comes from the second source (same row stride: host)
    const unsigned lda2 = (unsigned)(p.lda0 * 2);

    float* const blds = (float*)(smem + PANEL);                    // the panel's bias, one float per W row
    char* const wlds = smem + PANEL + ws_bias_bytes(NT) + wid * STG;   // this wave's staging slice
    // B fragment address inside a K-tile: row fr (+ 16 nt), 16-byte chunk (kh*4 + fg) ^ (fr & 7)
    const int b_rd[2] = {fr * 128 + ((fg ^ (fr & 7)) << 4), fr * 128 + (((4 + fg) ^ (fr & 7)) << 4)};

    // A fragments: lane (fr, fg) holds A[m_w + mt*16 + fr][kt*64 + kh*32 + fg*8 .. +8] in ring slot kt % RD
    half8 af[RD][2][MT];
    struct Offs { unsigned v[MT]; };
    auto tile_offsets = [&](int m_w, bool valid) {
        Offs o;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m_w + mt * 16 + fr;
            o.v[mt] = (valid && m < p.M) ? (unsigned)m * lda2 + (unsigned)(fg * 16) : OOB;
        }
        return o;
    };
    auto load_a = [&](int s, int kh, int kt, const Offs& o) {
        const bool sec = kt >= kt_second;                           // wave-uniform: scalar selects
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(sec ? p.A1 : p.A0), 0, OOB, RSRC_FLAGS);
        const int so = (sec ? kt - kt_second : kt) * 128 + kh * 64;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[s][kh][mt] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rA, o.v[mt], so, 0));
    };

#ifdef SDE_DIAG
    // diagnostic build: wall-clock stamps (s_memrealtime, 10 ns ticks) of this wave's milestones -> p.stamps[block][wave][16]
    unsigned long long st_t[16];
    int st_n = 0;
#define WS_STAMP() do { if (st_n < 16) st_t[st_n++] = __builtin_amdgcn_s_memrealtime(); } while (0)
    WS_STAMP();
#else
#define WS_STAMP() do { } while (0)
#endif
    for (int pass = 0; pass < npass; ++pass) {
        const int panel = pass * npp + pin;
        const bool work = active && panel < np;
        const int n0 = panel * PN;
        int c = c_lo + slot;
        bool have = work && c < c_hi;
        Offs oc = tile_offsets(c * CHUNK + wid * TROWS, have);
        // the first tile's A requests go out ahead of the panel's
#pragma unroll
        for (int s = 0; s < RD; ++s) { load_a(s, 0, s, oc); load_a(s, 1, s, oc); }
        if (work) {
            // W panel: K-tile kt, rows 8 r8 .. 8 r8 + 7 are one 1-KiB piece (lane -> row lane >> 3, source chunk (lane & 7) ^ row)
            const int srow = lane >> 3;
            const unsigned lchunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
            constexpr int npieces = KT * (PN / 8);
            for (int piece = wid; piece < npieces; piece += WAVES) {
                const int kt = piece / (PN / 8), r8 = piece - kt * (PN / 8);
                const unsigned voff = (unsigned)((n0 + r8 * 8 + srow) * p.K + kt * 64) * 2u + lchunk16;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(smem + kt * (PN * 128) + r8 * 1024), 16, voff, 0, 0, 0);
            }
            if (tid < PN) {
                const int n = n0 + tid;
                float b = 0.f;
                if (p.bias) {
                    if (EPI == WS_EPI_GEGLU) {   // W rows in fours: (value 2q, value 2q+1, gate 2q, gate 2q+1)
                        const int half_n = p.N >> 1, j = ((n >> 2) << 1) + (n & 1);
                        if (j < half_n) b = p.bias[((n & 2) ? half_n : 0) + j];
                    } else if (n < p.N) b = p.bias[n];
                }
                blds[tid] = b;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the panel and the first tile's fragments have landed
        __syncthreads();
        WS_STAMP();

        while (have) {
            const int m_w = c * CHUNK + wid * TROWS;
            const int cn = c + slots;
            const bool next = cn < c_hi;
            const Offs on = tile_offsets(cn * CHUNK + wid * TROWS, next);   // no next tile: out-of-range offsets, the loads return zeros
            floatx4 acc[NT][MT];
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int s = kt % RD;
#pragma unroll
                for (int kh = 0; kh < 2; ++kh) {
                    const char* wk = smem + kt * (PN * 128) + b_rd[kh];
#pragma unroll
                    for (int h5 = 0; h5 < NT / 5; ++h5) {
                        half8 bf[5];
#pragma unroll
                        for (int i = 0; i < 5; ++i) bf[i] = *(const half8*)(wk + (h5 * 5 + i) * 2048);
#pragma unroll
                        for (int i = 0; i < 5; ++i)
#pragma unroll
                            for (int mt = 0; mt < MT; ++mt)
                                acc[h5 * 5 + i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[i], af[s][kh][mt], acc[h5 * 5 + i][mt], 0, 0, 0);
                    }
                    // the registers just read take the fragments five K-tiles further down the stream
                    load_a(s, kh, kt, on);
                }
            }
            WS_STAMP();

            // ---- epilogue: acc[nt][mt][r] <-> n = n0 + nt*16 + 4*fg + r ; m = m_w + mt*16 + fr
            if constexpr (EPI == WS_EPI_GEGLU) {
                constexpr int PITCH = NT * 16 + 16;          // fp16 staging row: 8 NT output columns
                constexpr int CH = NT;                        // 16-byte chunks per row
                constexpr int NCH = 16 * CH, NI = (NCH + 63) / 64;
                const int half_n = p.N >> 1, j0 = n0 >> 1;
                auto rows = [&](auto tanh_tag) {
                    constexpr bool TANH = decltype(tanh_tag)::value;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const floatx4 b = *(const floatx4*)(blds + nt * 16 + 4 * fg);
                            const float g0 = acc[nt][mt][2] + b[2], g1 = acc[nt][mt][3] + b[3];
                            const half2_t h = {(half_t)((acc[nt][mt][0] + b[0]) * (TANH ? gelu_tanh_form(g0) : gelu_erf(g0))),
                                               (half_t)((acc[nt][mt][1] + b[1]) * (TANH ? gelu_tanh_form(g1) : gelu_erf(g1)))};
                            *(half2_t*)(wlds + fr * PITCH + (nt * 8 + 2 * fg) * 2) = h;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const int id = lane + 64 * i, idc = min(id, NCH - 1), row = idc / CH, ch = idc - row * CH;
                            const int m = m_w + mt * 16 + row, j = j0 + ch * 8;
                            const uint4v v = *(const uint4v*)(wlds + row * PITCH + ch * 16);
                            const unsigned off = (id < NCH && m < p.M && j < half_n) ? (unsigned)(m * p.ldd + j) * 2u : OOB;
                            __builtin_amdgcn_raw_buffer_store_b128(v, rD, off, 0, 0);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                };
                if (p.gelu_tanh) rows(std::true_type{}); else rows(std::false_type{});
            } else if constexpr (EPI == WS_EPI_H16) {
                // no residual: the result is rounded once either way, so the staging rows hold fp16 (half the LDS of the fp32 form,
                // which is what lets sixteen waves share the CU beside the panel)
                constexpr int PITCH = WS_STG16_PITCH, CH = 10, NCH = 16 * CH, NI = 3;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int cg = 0; cg < NT / 5; ++cg) {
#pragma unroll
                        for (int i5 = 0; i5 < 5; ++i5) {
                            const int nt = cg * 5 + i5;
                            const floatx4 b = *(const floatx4*)(blds + nt * 16 + 4 * fg);
                            half4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[nt][mt][r] * p.alpha + b[r]);
                            *(half4*)(wlds + fr * PITCH + (i5 * 16 + 4 * fg) * 2) = o;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const int id = lane + 64 * i, idc = min(id, NCH - 1), row = idc / CH, ch = idc - row * CH;
                            const int m = m_w + mt * 16 + row, n = n0 + cg * 80 + ch * 8;
                            const uint4v v = *(const uint4v*)(wlds + row * PITCH + ch * 16);
                            const unsigned off = (id < NCH && m < p.M && n < p.N) ? (unsigned)(m * p.ldd + n) * 2u : OOB;
                            __builtin_amdgcn_raw_buffer_store_b128(v, rD, off, 0, 0);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            } else {
                constexpr bool LO = EPI == WS_EPI_LO;
                constexpr int PITCH = WS_STG_PITCH, CH = 10;  // 80 columns per step: ten 8-column chunks per row
                constexpr int NCH = 16 * CH, NI = 3;
                const bool haslo = LO && p.Rlo, dlo = LO && p.Dlo;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int cg = 0; cg < NT / 5; ++cg) {
                        half8 rr[HAS_R ? NI : 1], rl[(HAS_R && LO) ? NI : 1];
                        if constexpr (HAS_R) {
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
                                const int id = min(lane + 64 * i, NCH - 1), row = id / CH, ch = id - row * CH;
                                const int m = min(m_w + mt * 16 + row, p.M - 1), n = min(n0 + cg * 80 + ch * 8, p.N - 8);
                                const unsigned off = (unsigned)(m * p.ldr + n) * 2u;
                                rr[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rR, off, 0, 0));
                                if constexpr (LO) rl[i] = __builtin_bit_cast(half8, __builtin_amdgcn_raw_buffer_load_b128(rRl, haslo ? off : OOB, 0, 0));
                            }
                        }
#pragma unroll
                        for (int i5 = 0; i5 < 5; ++i5) {
                            const int nt = cg * 5 + i5;
                            const floatx4 b = *(const floatx4*)(blds + nt * 16 + 4 * fg);
                            floatx4 o;
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[r] = acc[nt][mt][r] * p.alpha + b[r];
                            *(floatx4*)(wlds + fr * PITCH + (i5 * 16 + 4 * fg) * 4) = o;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            const int id = lane + 64 * i, idc = min(id, NCH - 1), row = idc / CH, ch = idc - row * CH;
                            const int m = m_w + mt * 16 + row, n = n0 + cg * 80 + ch * 8;
                            const floatx4 lo = *(const floatx4*)(wlds + row * PITCH + ch * 32), hi = *(const floatx4*)(wlds + row * PITCH + ch * 32 + 16);
                            const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            half8 v, vl;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                float x = f[e];
                                if constexpr (HAS_R) x += (float)rr[i][e];
                                if constexpr (HAS_R && LO) x += (float)rl[i][e];   // no low part: the load was out of range = zeros
                                v[e] = (half_t)x;
                                if constexpr (LO) vl[e] = (half_t)(x - (float)v[e]);
                            }
                            const unsigned off = (id < NCH && m < p.M && n < p.N) ? (unsigned)(m * p.ldd + n) * 2u : OOB;
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, v), rD, off, 0, 0);
                            if constexpr (LO) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v, vl), rDl, dlo ? off : OOB, 0, 0);
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            WS_STAMP();
            oc = on; c = cn; have = next;
        }
        __syncthreads();   // every wave is done with this panel before the next pass overwrites it
    }
#ifdef SDE_DIAG
    if (p.stamps && lane == 0) {
        unsigned long long* dst = (unsigned long long*)p.stamps + ((size_t)blockIdx.x * 16 + wid) * 16;
        for (int i = 0; i < 16; ++i) dst[i] = i < st_n ? st_t[i] : 0ull;
    }
#endif
#undef WS_STAMP
}

// ------------------------------------------------------------------------------------------ host side
bool gemm_mfma_supported(const GemmArgs& a);

static bool ws_k_ok(const GemmArgs& a) { return a.K == 320; }
// panel width: 160 columns (80 for the split-fp16 epilogue, whose registers do not fit beside 80 accumulators at 32 rows per tile)
static int ws_panel_cols(int K, bool lo) { (void)K; return lo ? 80 : 160; }

bool gemm_ws_supported(const GemmArgs& a)
{
    if (a.taps != 1 || a.batch != 1 || (a.epi != EPI_STD && a.epi != EPI_GEGLU_PAIR)) return false;
    if (a.K != a.C0 + a.C1 || !ws_k_ok(a)) return false;
    if (a.A1 && (a.C0 % 64 || a.lda1 != a.lda0)) return false;      // both sources of a [hi | lo] operand share one row stride
    if (a.lda0 % 8 || a.ldd % 8 || (a.N & 7) || a.Npad % 160) return false;
    if (a.R && (a.ldr % 8)) return false;
    if (a.epi == EPI_GEGLU_PAIR && ((a.N >> 1) & 7 || a.R || a.Rlo || a.Dlo)) return false;
    // 32-bit byte offsets inside the descriptor windows
    if ((long long)a.M * a.lda0 * 2 >= 0x7ff00000LL || (long long)a.M * a.ldd * 2 >= 0x7ff00000LL || (a.R && (long long)a.M * a.ldr * 2 >= 0x7ff00000LL)) return false;
    if ((long long)a.Npad * a.K * 2 >= 0x7ff00000LL) return false;
    return true;
}

// panels per pass: cost model in units of one 256-row chunk of one panel; a W reload costs ~0.25 of that (100 KB out of L2)
static int ws_pick_npp(int np, int G, int chunks_per_xcd)
{
    int best = 1; double best_cost = 1e30;
    for (int npp = 1; npp <= std::min(np, G); ++npp) {
        const int slots = G / npp, passes = (np + npp - 1) / npp;
        const int rounds = (chunks_per_xcd + slots - 1) / slots;
        // every pass streams all of A again, out of HBM (measured: N = 960 in three passes of two panels fetched A three times)
        const double cost = passes * (rounds + 0.25) + (passes - 1) * 0.5 * chunks_per_xcd / std::max(1, slots);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = npp; }
    }
    return best;
}

// rows per wave tile: 32 for the GEGLU projection (MFMA / VALU-issue bound: half the LDS reads per MFMA), 16 for the streaming epilogues
// (HBM bound: twice the tiles per wave = half the pipeline's fill and drain)
static int ws_mt(const GemmDev& p)
{
    static const int mt_env = getenv("SDE_WS_MT") ? atoi(getenv("SDE_WS_MT")) : 0;   // tuning hook
    if (mt_env == 1 || mt_env == 2) return mt_env;
    return p.epi == EPI_GEGLU_PAIR ? 2 : 1;
}

template <int KT, int NT, int MT>
static void launch_ws_m(const LaunchCtx& ctx, const GemmDev& p, dim3 grid)
{
    constexpr int smem = ws_smem_bytes(KT, NT);
    static_assert(smem <= 160 * 1024, "LDS");
    auto go = [&](auto kern) {
        allow_dynamic_lds((const void*)kern, smem);
        hipLaunchKernelGGL(kern, grid, dim3(512), smem, ctx.stream, p);
    };
    if (p.Rlo || p.Dlo) {
        // the split-fp16 epilogue beside 80 accumulators does not fit 256 registers: launch_gemm_ws sends it to 80-column panels
        if constexpr (NT == 5) { if (p.R) go(gemm_ws_kernel<KT, NT, MT, WS_EPI_LO, true>); else go(gemm_ws_kernel<KT, NT, MT, WS_EPI_LO, false>); }
        else SDE_REQUIRE(false, SDE_ERR_INVALID_ARGUMENT, "gemm_ws: split-fp16 epilogue needs 80-column panels");
    } else if (p.epi == EPI_GEGLU_PAIR) go(gemm_ws_kernel<KT, NT, MT, WS_EPI_GEGLU, false>);
    else if (p.R) go(gemm_ws_kernel<KT, NT, MT, WS_EPI_STD, true>);
    else {
        static const int w16_env = getenv("SDE_WS_W16") ? atoi(getenv("SDE_WS_W16")) : 1;   // tuning hook: 0 = eight waves, fp32 staging
        if constexpr (MT == 1) {
            if (w16_env) {
                // sixteen waves (<= 128 registers each, fp16 staging): twice the requests in flight for the HBM-bound projections
                constexpr int smem16 = ws_smem_bytes(KT, NT, 16, WS_STG16_BYTES);
                static_assert(smem16 <= 160 * 1024, "LDS");
                auto kern = gemm_ws_kernel<KT, NT, 1, WS_EPI_H16, false, 16>;
                allow_dynamic_lds((const void*)kern, smem16);
                hipLaunchKernelGGL(kern, grid, dim3(1024), smem16, ctx.stream, p);
                return;
            }
        }
        go(gemm_ws_kernel<KT, NT, MT, WS_EPI_STD, false>);
    }
}
template <int KT, int NT>
static void launch_ws_t(const LaunchCtx& ctx, const GemmDev& p, dim3 grid)
{
    if (ws_mt(p) == 2) launch_ws_m<KT, NT, 2>(ctx, p, grid); else launch_ws_m<KT, NT, 1>(ctx, p, grid);
}

void launch_gemm_ws(const LaunchCtx& ctx, GemmDev& p)
{
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    const int G = std::max(1, cus / 8);
    const bool lo = p.Rlo || p.Dlo;
    const int PN = ws_panel_cols(p.K, lo), np = p.Npad / PN;
    const bool w16 = ws_mt(p) == 1 && !lo && !p.R && p.epi == EPI_STD && !(getenv("SDE_WS_W16") && !atoi(getenv("SDE_WS_W16")));
    const int chunk = (w16 ? 16 : WS_WAVES) * 16 * ws_mt(p), nchunks = (p.M + chunk - 1) / chunk;
    static const int npp_env = getenv("SDE_WS_NPP") ? atoi(getenv("SDE_WS_NPP")) : 0;   // tuning hook
    p.ws_npp = npp_env > 0 ? std::min(std::min(npp_env, np), G) : ws_pick_npp(np, G, (nchunks + 7) / 8);
    p.splitk = 1; p.ws = nullptr;
    dim3 grid(8 * G, 1, 1);
#ifdef SDE_DIAG
    static const bool stamps = getenv("SDE_STAMPS") && atoi(getenv("SDE_STAMPS"));
    if (stamps && ctx.splitk_ws) { p.stamps = (unsigned*)ctx.splitk_ws; HIP_CHECK(hipMemsetAsync(p.stamps, 0, (size_t)grid.x * 16 * 16 * 8, ctx.stream)); }
#endif
    if (PN == 160) launch_ws_t<5, 10>(ctx, p, grid);
    else launch_ws_t<5, 5>(ctx, p, grid);
#ifdef SDE_DIAG
    if (p.stamps) {
        std::vector<unsigned long long> h((size_t)grid.x * 16 * 16);
        HIP_CHECK(hipStreamSynchronize(ctx.stream));
        HIP_CHECK(hipMemcpy(h.data(), p.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (auto v : h) if (v) { t0 = std::min(t0, v); t1 = std::max(t1, v); }
        fprintf(stderr, "[ws stamps] M=%d N=%d K=%d npp=%d: first stamp -> last stamp %.2f us; per wave (us since first stamp): start | panel ready | (K loop end, epilogue end) per tile\n",
                p.M, p.N, p.K, p.ws_npp, (t1 - t0) * 0.01);
        for (int b : {0, 1, 8, 9, 100, 255}) {
            if (b >= (int)grid.x) continue;
            for (int w : {0, 4, 7}) {
                fprintf(stderr, "  block %3d wave %d:", b, w);
                for (int i = 0; i < 16; ++i) { unsigned long long v = h[((size_t)b * 16 + w) * 16 + i]; if (v) fprintf(stderr, " %6.2f", (v - t0) * 0.01); }
                fprintf(stderr, "\n");
            }
        }
        // distribution of the waves' end times
        std::vector<double> ends;
        for (size_t i = 0; i < h.size() / 16; ++i) { unsigned long long e = 0; for (int k = 0; k < 16; ++k) e = std::max(e, h[i * 16 + k]); if (e) ends.push_back((e - t0) * 0.01); }
        std::sort(ends.begin(), ends.end());
        if (!ends.empty()) fprintf(stderr, "  wave end times: min %.2f  median %.2f  p90 %.2f  max %.2f us (%zu waves)\n", ends.front(), ends[ends.size() / 2], ends[ends.size() * 9 / 10], ends.back(), ends.size());
    }
#endif
}
