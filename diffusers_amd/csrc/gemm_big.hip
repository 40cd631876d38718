// gemm_big.hip — 8-wave 256x320x64 "ping-pong" fp16 GEMM / implicit-GEMM convolution for gfx950 (MI355X).
//
// Same contract as gemm.hip (D = epilogue(A * W^T), A through an implicit im2col), for the large MFMA-bound shapes
// of the UNet (3x3 convs and wide linears with N % 320 == 0).  Why a second kernel: with 128x160 tiles the
// global->LDS path (64 B/clk/CU) is ~90 % busy per MFMA-bound K step; a 256x320 tile brings that to 45 %.
//
// Structure:
//   * 512 threads = 8 waves as 2 (M) x 4 (N); each wave owns a 128 x 80 sub-tile = 8 x 5 MFMA 16x16x32 tiles
//     (160 accumulator VGPRs), issued as D^T = W * A^T so a lane holds 4 consecutive n of one m.
//   * LDS: 2 K-tile buffers x (256 + 320) rows x 128 B = 144 KiB, filled by buffer_load_dwordx4 ... lds
//     (lane-linear image, XOR swizzle on the source chunk, same XOR on the ds_read_b128 address).  Padded taps and
//     tail rows use the buffer descriptor's range check (offset >= num_records returns zeros): no zero page, no
//     64-bit address arithmetic.
//   * Each K-tile is 4 phases (m-half x k-half) of {ds_read fragments, issue loads | barrier | 20 MFMA | barrier}.
//     Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA section while its
//     partner is in its load section.  Loads for K-tile t+1 are spread over the sections of tile t and retired with
//     one s_waitcnt vmcnt(0) per tile placed at least one full section after the last issue (never right behind an issue).
//   * XCD-aware bijective block remap, optional split-K (partial fp32 slabs reduced by gemm.hip's reduce kernel).
#include "gemm_common.h"

#include <algorithm>
#include <type_traits>

#ifndef BIG_RG
#define BIG_RG(LO) ((LO) ? 1 : 2)
#endif
namespace {
constexpr int BM = 256;
constexpr int A_BYTES = BM * 128;
constexpr int NPA = BM / 64;                                                        // 64-row staging parts of the A tile
constexpr int smem_bytes(int nt) { return 2 * (BM + nt * 64) * 128; }               // NT = 5: 147,456 B ; NT = 4: 131,072 B
constexpr unsigned OOB = 0x80000000u;                                               // >= num_records -> reads as zero
constexpr int RSRC_FLAGS = 0x00020000;

}  // namespace

// VAR: 0 in the product.  Diagnostic builds (-DSDE_DIAG, wrong results, timing only): bit0 = no loads in the main loop,
// bit1 = no fragment reads, bit2 = no barriers.
// NT = 16-column tiles per wave: 5 -> 256x320 block tile (UNet widths 320/640/1280), 4 -> 256x256 (VAE widths 256/512).
// LO: the split-fp16 epilogue (residual and / or result as hi + lo pairs, kernels.h).  A separate instance because its extra live
// registers (second residual set, low halves) push the 256-VGPR kernel into scratch spills inside the latency-critical epilogue:
// compiled into the plain instance it cost the 64x64-level linears 25-30 % (27.9 -> 35.3 us at M=65536, N=K=320).
template <bool CONV, bool UPS, int NT, int VAR = 0, bool LO = false>
__global__ __launch_bounds__(512) void gemm_big_kernel(GemmDev p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BN = NT * 64, WN = NT * 16;                       // block / wave tile width
    constexpr int STAGE = A_BYTES + BN * 128;                       // bytes per K-tile buffer
    constexpr int NPART = NPA + NT;                                 // staging parts per K-tile (A: 4, W: NT)
    // loads issued per section for the next K-tile; sections are L0 M0 L1 M1 L2 M2 L3 M3 (L = load, M = MFMA section).
    // Waves 0-3 may not touch the other buffer before M0 (waves 4-7 are still reading it during L0).
    // Measured (tools/microbench.py, MI355X): placements differ by +-3 %; the cost is the load path's throughput, not the slot.
    constexpr int kScheds[4][2][8] = {
        {{0, 0, NT, 0, 4, 0, 0, 0}, {4, 0, NT, 0, 0, 0, 0, 0}},   // product schedule: load sections only, early in the tile
        {{0, 2, 2, 1, 2, NT - 3, 0, 0}, {3, 1, 2, 1, NT - 3, 0, 0, 0}},   // diagnostic alternatives (VAR >> 4, -DSDE_DIAG builds)
        {{0, 3, 0, 3, 0, NT - 2, 0, 0}, {0, 3, 0, 3, 0, NT - 2, 0, 0}},
        {{0, 1, 3, 1, 2, 0, NT - 3, 0}, {3, 1, 3, 0, NT - 3, 0, 0, 0}},
    };
    static_assert(kScheds[VAR >> 4][0][0] == 0 && kScheds[VAR >> 4][1][6] == 0 && kScheds[VAR >> 4][0][7] == 0 && kScheds[VAR >> 4][1][7] == 0, "illegal issue slot");
#define kSched kScheds[VAR >> 4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int z = blockIdx.z;
    const int fr = lane & 15, fg = lane >> 4;

    // XCD-aware bijective remap (blocks b, b+8, ... share an XCD/L2): each XCD gets a contiguous range of tiles
    const int nwg = p.nbm * p.nbn;
    int bid = blockIdx.x;
    {
        int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // each XCD re-reads whatever its tiles share: keep the larger operand's panel behind one L2 (host picks, launch_gemm)
    const int tm = p.colmajor ? bid % p.nbm : bid / p.nbn, tn = p.colmajor ? bid / p.nbm : bid - (bid / p.nbn) * p.nbn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk_all = p.K >> 6;
    const int ks = blockIdx.y;
    const int kt0 = (int)((long long)nk_all * ks / p.splitk);
    const int nk = (int)((long long)nk_all * (ks + 1) / p.splitk) - kt0;

    // buffer descriptors (raw, 2 GiB window; padded taps / tail rows use offset OOB)
    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A0 + z * p.sA), 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)((p.A1 ? p.A1 : p.A0) + z * p.sA), 0, OOB, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + z * p.sW), 0, OOB, RSRC_FLAGS);

    // ---- staging geometry: part j covers rows [64j, 64j+64) of the A tile (j < 4) or of the W tile (j >= 4);
    //      wave w stages rows 8w..8w+7 of the part, lane -> (row = lane>>3, 16-B chunk = (lane&7) ^ row)
    const int srow = lane >> 3;
    const unsigned lchunk16 = (unsigned)(((lane & 7) ^ srow) << 4);
    unsigned a_off0[NPA], a_off1[NPA];   // byte offset of the window-centre pixel (conv) / row (linear) + swizzled chunk, per source
    unsigned a_mask[NPA];                // bit t: tap t in bounds (and row < M); bits 9,10: oy,ox parity (fused 2x upsample)
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int m = m0 + j * 64 + wid * 8 + srow;
        const bool ok = m < p.M;
        int pix = m;
        if (CONV) {
            const int hw = p.Ho * p.Wo;
            const int nb = m / hw, rem = m - nb * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            const int cy = oy * p.stride + p.coff, cx = ox * p.stride + p.coff;   // window centre in the (virtual) input
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
    const int w_part = 128 * p.K;   // byte stride between 64-row W parts

    // staging state of the K-tile currently being issued (wave-uniform scalars, recomputed once per K-tile)
    // K order (kernels.h): conv K-tiles run chunk-major, kt = chunk*9 + tap, so consecutive tiles re-read the same lines
    int st_k0 = kt0 << 6, st_tap = 0, st_c = st_k0;
    if (CONV) { st_tap = kt0 % 9; st_c = (kt0 / 9) << 6; }
    bool st_second = false;      // the K-tile's channels come from the second source of a concat
    unsigned st_bit = 1u;        // 1 << tap
    int st_add = 0;              // byte offset of (tap, channel) relative to the window centre (not UPS: includes the tap shift)
    int st_ld2 = 0, st_ey0 = 0, st_ey1 = 0, st_ex0 = 0, st_ex1 = 0;   // UPS only: per-parity tap shifts
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
    // issue staging part j of the current staging K-tile into buffer `buf`
    auto issue = [&](int j, int buf) {
        char* dst = smem + buf * STAGE + (j < NPA ? j * 8192 : A_BYTES + (j - NPA) * 8192) + wid * 1024;
        if (VAR & 8) {   // diagnostic: same instruction stream to the memory system, no address arithmetic
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(dst), 16, w_off, 0, 0, 0);
        } else if (j < NPA) {
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
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(dst), 16, w_off, (j - NPA) * w_part + st_k0 * 2, 0, 0);
        }
    };

    // ---- fragment read addresses (swizzled): A rows wr*128 + mt*16 + fr, W rows wc*WN + nt*16 + fr
    const int swz0 = ((fg) ^ (fr & 7)) << 4, swz1 = ((4 + fg) ^ (fr & 7)) << 4;
    const int a_rd0 = (wr * 128 + fr) * 128 + swz0, a_rd1 = (wr * 128 + fr) * 128 + swz1;
    const int b_rd0 = A_BYTES + (wc * WN + fr) * 128 + swz0, b_rd1 = A_BYTES + (wc * WN + fr) * 128 + swz1;

    floatx4 acc[NT][8];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    half8 bf[NT], af[4];
    if (VAR & 2) {
#pragma unroll
        for (int i = 0; i < NT; ++i) bf[i] = *(const half8*)(smem + lane * 16 + i * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const half8*)(smem + lane * 16 + 8192 + i * 1024);
    }

    // ---- prologue: K-tile 0 into buffer 0
#pragma unroll
    for (int j = 0; j < NPART; ++j) issue(j, 0);
    stage_advance();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one barrier behind

#define LOAD_B(ofs)                                                                                   \
    if (!(VAR & 2)) _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) bf[nt] = *(const half8*)(cur + (ofs) + nt * 2048);
#define LOAD_A(ofs, mh)                                                                               \
    if (!(VAR & 2)) _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) af[mt] = *(const half8*)(cur + (ofs) + ((mh) * 4 + mt) * 2048);
#define ISSUE_N(sec)                                                                                  \
    if (more && !(VAR & 1)) {                                                                         \
        _Pragma("unroll") for (int i = 0; i < kSched[0][sec]; ++i) if (wr == 0) issue(jn0 + i, nb_);   \
        _Pragma("unroll") for (int i = 0; i < kSched[1][sec]; ++i) if (wr == 1) issue(jn1 + i, nb_);   \
    }                                                                                                 \
    jn0 += kSched[0][sec]; jn1 += kSched[1][sec];
#define MFMA_BLOCK(mh)                                                                                \
    __builtin_amdgcn_s_setprio(1);                                                                    \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                                                  \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                              \
            acc[nt][(mh) * 4 + mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[nt], af[mt], acc[nt][(mh) * 4 + mt], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
#define FENCE() __builtin_amdgcn_sched_barrier(0)

#define BAR() do { if (!(VAR & 4)) __builtin_amdgcn_s_barrier(); } while (0)
    for (int kt = 0; kt < nk; ++kt) {
        const char* cur = smem + (kt & 1) * STAGE;
        const int nb_ = (kt & 1) ^ 1;
        const bool more = kt + 1 < nk;
        int jn0 = 0, jn1 = 0;
        // phase 0: (m-half 0, k-half 0)
        LOAD_B(b_rd0) LOAD_A(a_rd0, 0)
        ISSUE_N(0)
        FENCE(); BAR(); FENCE();
        ISSUE_N(1)
        MFMA_BLOCK(0)
        FENCE(); BAR(); FENCE();
        // phase 1: (m-half 1, k-half 0)
        LOAD_A(a_rd0, 1)
        ISSUE_N(2)
        FENCE(); BAR(); FENCE();
        ISSUE_N(3)
        MFMA_BLOCK(1)
        FENCE(); BAR(); FENCE();
        // phase 2: (m-half 1, k-half 1)
        LOAD_B(b_rd1) LOAD_A(a_rd1, 1)
        ISSUE_N(4)
        FENCE(); BAR(); FENCE();
        ISSUE_N(5)
        MFMA_BLOCK(1)
        FENCE(); BAR(); FENCE();
        // phase 3: (m-half 0, k-half 1)
        LOAD_A(a_rd1, 0)
        ISSUE_N(6)   // waves 0-3 only (waves 4-7 wait for their loads right below)
        FENCE();
        if (wr == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // waves 4-7: last issue was in L2, two sections ago
        BAR(); FENCE();
        if (more) stage_advance();   // the next tile's scalar address work rides under the MFMAs (all of this tile's issues are done)
        MFMA_BLOCK(0)
        FENCE();
        if (wr == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // waves 0-3: last issue at the latest in L3, one MFMA section ago
        BAR(); FENCE();
    }
#undef BAR
    if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the extra barrier of waves 4-7
#undef kSched
#undef LOAD_B
#undef LOAD_A
#undef ISSUE_N
#undef MFMA_BLOCK
#undef FENCE

    // ---- epilogue: acc[nt][mt][r] <-> n = n0 + wc*WN + nt*16 + 4*fg + r ; m = m0 + wr*128 + mt*16 + fr
    const long long zD = z * p.sD, zR = z * p.sR;
    if (p.splitk > 1) {
        float* ws = p.ws + (long long)ks * p.M * p.Npad;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const int m = m0 + wr * 128 + mt * 16 + fr;
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
    // Results go through the wave's own 18 KiB slice of the (now idle) staging LDS so that global stores are whole 16-byte
    // chunks in row order (80-160 B runs per row) instead of 4/8-byte pieces scattered over 16 rows per instruction.
    // The slice is private to the wave: only wave-level ordering is needed (s_waitcnt + wave_barrier), no s_barrier.
    char* const wlds = smem + wid * (smem_bytes(NT) / 8);
    const bool vec_ok = (p.ldd & 7) == 0;   // 16-byte aligned rows (host guarantees N % 8 == 0 on this path)
    if (p.epi == EPI_GEGLU_PAIR) {
        // weight rows interleaved in fours: a lane's 4 consecutive n are (value_j, value_j+1, gate_j, gate_j+1), j = n/2: both values and
        // both gates are adjacent accumulator registers, so the packed fp32 math below needs no register moves to pair them up
        // (with (value, gate, value, gate) the compiler spent two v_mov per GELU on that)
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
        constexpr int OW = WN / 2;              // output columns per wave
        constexpr int PITCH = OW * 2 + 16;      // LDS row pitch (bytes)
        constexpr int CH = OW * 2 / 16;         // 16-byte chunks per row
        const int half_n = p.N >> 1;
        const int j0 = (n0 + wc * WN) >> 1;
        float bvv[NT][2], bvg[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int j = j0 + nt * 8 + 2 * fg;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool ok = p.bias && j + r < half_n;
                bvv[nt][r] = ok ? p.bias[j + r] : 0.f;
                bvg[nt][r] = ok ? p.bias[half_n + j + r] : 0.f;
            }
        }
        half_t* D = p.D + zD;
        // the GELU form is chosen ONCE: with the test inside gelu_sel every value sat behind its own branch, and the rcp / exp latencies of
        // one value could not be covered by the arithmetic of the next
        auto geglu_rows = [&](auto tanh_tag) {
        constexpr bool TANH = decltype(tanh_tag)::value;
#pragma unroll
        for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
            for (int mq = 0; mq < 4; ++mq) {
                const int mt = mh * 4 + mq;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const half2_t h = {(half_t)((acc[nt][mt][0] + bvv[nt][0]) * (TANH ? gelu_tanh_form(acc[nt][mt][2] + bvg[nt][0]) : gelu_erf(acc[nt][mt][2] + bvg[nt][0]))),
                                       (half_t)((acc[nt][mt][1] + bvv[nt][1]) * (TANH ? gelu_tanh_form(acc[nt][mt][3] + bvg[nt][1]) : gelu_erf(acc[nt][mt][3] + bvg[nt][1])))};
                    *(half2_t*)(wlds + (mq * 16 + fr) * PITCH + (nt * 8 + 2 * fg) * 2) = h;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int id = lane + 64 * i, row = id / CH, ch = id - row * CH;
                const int m = m0 + wr * 128 + mh * 64 + row, j = j0 + ch * 8;
                const half8 v = *(const half8*)(wlds + row * PITCH + ch * 16);
                if (m < p.M && j < half_n) {
                    if (vec_ok && j + 8 <= half_n) *(half8*)(D + (long long)m * p.ldd + j) = v;
                    else for (int e = 0; e < 8 && j + e < half_n; ++e) D[(long long)m * p.ldd + j + e] = v[e];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        };
        if (p.gelu_tanh) geglu_rows(std::true_type{}); else geglu_rows(std::false_type{});
        return;
    }
    {   // EPI_STD, 32 rows (two row groups) at a time: fp32 in LDS so the residual is added before the single rounding.
        // The split-fp16 instance goes 16 rows at a time: its second residual set and low halves would not fit the register file otherwise.
        constexpr int PITCH = WN * 4 + 32;      // LDS row pitch (bytes, fp32)
        constexpr int CH = WN / 8;              // 8-column chunks per row (32 B in LDS, 16 B in memory)
        constexpr int RG = BIG_RG(LO);          // 16-row groups per step
        constexpr int NCH = RG * 16 * CH;       // chunks per step
        constexpr int NI = (NCH + 63) / 64;     // chunks per lane and step
        float bv[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = n0 + wc * WN + nt * 16 + 4 * fg;
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[nt][r] = (p.bias && n < p.N) ? p.bias[n + r] : 0.f;
        }
        const half_t* R = p.R ? p.R + zR : nullptr;
        const half_t* Rlo = (LO && p.R && p.Rlo) ? p.Rlo + zR : nullptr;   // split-fp16 residual / result (kernels.h)
        half_t* D = p.D + zD;
        half_t* Dlo = (LO && p.Dlo) ? p.Dlo + zD : nullptr;
        const bool rvec_ok = R && (p.ldr & 7) == 0;
        if constexpr (LO) {
            if (R && rvec_ok && vec_ok && (p.N & 7) == 0) {
                // Split-fp16 residual through LDS, two steps ahead.  Loaded into registers per 16-row step (below) every step pays an HBM
                // round trip in series: eight per wave, with 656 KB per CU to move (R hi + lo in, D hi + lo out) the epilogue of the
                // K = 640 shortcut / proj_out GEMMs ran at a third of the HBM rate.  Here each lane's chunks of step s+2 go global -> LDS by
                // buffer_load ... lds into a two-slot ring behind the fp32 staging rows of the wave's slice while step s is added, split
                // and stored.  A lane reads back exactly the chunks it fetched (same id -> (row, chunk) map), so the only ordering needed
                // is its own counted vmcnt: the queue behind R(s) holds the stores of step s-1 (a wave-uniformly skipped store only
                // shortens it) and the K pieces of R(s+1), so "at most K outstanding" means R(s) has landed.
                constexpr int STG = 16 * PITCH, SLOT = 2 * NI * 1024;
                static_assert(RG == 1 && STG + 2 * SLOT <= smem_bytes(NT) / 8, "residual ring must fit the wave's LDS slice");
                const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, OOB, RSRC_FLAGS);
                const __amdgpu_buffer_rsrc_t rRl = __builtin_amdgcn_make_buffer_rsrc((void*)(Rlo ? Rlo : R), 0, OOB, RSRC_FLAGS);
                const bool haslo = Rlo != nullptr;
                auto issue_R = [&](int ms, int slot) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int id = min(lane + 64 * i, NCH - 1), row = id / CH, ch = id - row * CH;
                        const int m = min(m0 + wr * 128 + ms * 16 + row, p.M - 1), n = min(n0 + wc * WN + ch * 8, p.N - 8);
                        const unsigned voff = (unsigned)(m * p.ldr + n) * 2u;
                        char* dst = wlds + STG + slot * SLOT + i * 1024;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rR, LDS_PTR(dst), 16, voff, 0, 0, 0);
                        if (haslo) __builtin_amdgcn_raw_ptr_buffer_load_lds(rRl, LDS_PTR(dst + NI * 1024), 16, voff, 0, 0, 0);
                    }
                };
                issue_R(0, 0);
                issue_R(1, 1);
#pragma unroll
                for (int ms = 0; ms < 8; ++ms) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        floatx4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = acc[nt][ms][r] * p.alpha + bv[nt][r];
                        *(floatx4*)(wlds + fr * PITCH + (nt * 16 + 4 * fg) * 4) = o;
                    }
                    if (ms == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (haslo) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    static_assert(NI == 3, "counted waits above assume three pieces per tensor and step");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    const char* slot = wlds + STG + (ms & 1) * SLOT;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        const int id = lane + 64 * i, idc = min(id, NCH - 1), row = idc / CH, ch = idc - row * CH;
                        const int m = m0 + wr * 128 + ms * 16 + row, n = n0 + wc * WN + ch * 8;
                        const floatx4 lo = *(const floatx4*)(wlds + row * PITCH + ch * 32), hi = *(const floatx4*)(wlds + row * PITCH + ch * 32 + 16);
                        const half8 rr = *(const half8*)(slot + i * 1024 + lane * 16);
                        half8 rl;
                        if (haslo) rl = *(const half8*)(slot + (NI + i) * 1024 + lane * 16);
                        const float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        half8 v, vl;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float x = f[e] + (float)rr[e];
                            if (haslo) x += (float)rl[e];
                            v[e] = (half_t)x;
                            vl[e] = (half_t)(x - (float)v[e]);
                        }
                        if (id < NCH && m < p.M && n < p.N) {
                            *(half8*)(D + (long long)m * p.ldd + n) = v;
                            if (Dlo) *(half8*)(Dlo + (long long)m * p.ldd + n) = vl;
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (ms + 2 < 8) issue_R(ms + 2, ms & 1);
                }
                return;
            }
        }
#pragma unroll
        for (int ms = 0; ms < 8 / RG; ++ms) {
            // residual chunks of these rows, in flight while the accumulators go through LDS
            half8 rr[NI], rl[LO ? NI : 1];
            if (R) {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int id = min(lane + 64 * i, NCH - 1), row = id / CH, ch = id - row * CH;
                    const int m = min(m0 + wr * 128 + ms * (16 * RG) + row, p.M - 1), n = min(n0 + wc * WN + ch * 8, p.N - 8);
                    if (rvec_ok) rr[i] = *(const half8*)(R + (long long)m * p.ldr + n);
                    else for (int e = 0; e < 8; ++e) rr[i][e] = R[(long long)m * p.ldr + n + e];
                    if constexpr (LO) if (Rlo) {
                        if (rvec_ok) rl[i] = *(const half8*)(Rlo + (long long)m * p.ldr + n);
                        else for (int e = 0; e < 8; ++e) rl[i][e] = Rlo[(long long)m * p.ldr + n + e];
                    }
                }
            }
#pragma unroll
            for (int mq = 0; mq < RG; ++mq) {
                const int mt = ms * RG + mq;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    floatx4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = acc[nt][mt][r] * p.alpha + bv[nt][r];
                    *(floatx4*)(wlds + (mq * 16 + fr) * PITCH + (nt * 16 + 4 * fg) * 4) = o;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int id = lane + 64 * i, idc = min(id, NCH - 1), row = idc / CH, ch = idc - row * CH;
                const int m = m0 + wr * 128 + ms * (16 * RG) + row, n = n0 + wc * WN + ch * 8;
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
                    if (vec_ok && n + 8 <= p.N) {
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
}

// ------------------------------------------------------------------------------------------ host side
// block tile width for this problem: 320 where the padded width allows it, else 256, else 0 (not supported)
int gemm_big_bn(const GemmArgs& a) { return a.Npad % 320 == 0 ? 320 : (a.Npad % 256 == 0 ? 256 : 0); }

bool gemm_big_supported(const GemmArgs& a)
{
    GemmArgs c = a;
    if (c.epi == EPI_GEGLU_PAIR) { c.epi = EPI_TRANS; }   // shared operand/alignment rules without gemm.hip's own epilogue constraints
    if (!gemm_mfma_supported(c)) return false;
    if (!gemm_big_bn(a) || (a.epi != EPI_STD && a.epi != EPI_GEGLU_PAIR) || (a.N & 3)) return false;   // other epilogues stay on gemm.hip's kernel
    if (a.epi == EPI_GEGLU_PAIR && (a.ldd & 1)) return false;
    if ((a.Rlo || a.Dlo) && gemm_big_bn(a) != 320) return false;   // the split-fp16 epilogue instance exists for the 256x320 tile only
    // 32-bit byte offsets inside a 2 GiB descriptor window
    const long long rowsA = a.taps == 9 ? (long long)(a.M / (a.Ho * a.Wo)) * (a.ups ? ((a.Hi + 1) / 2) * ((a.Wi + 1) / 2) : a.Hi * a.Wi) : a.M;
    if (rowsA >= (1 << 24) || a.lda0 * 2 >= (1 << 23) || a.lda1 * 2 >= (1 << 23) || a.Wi >= (1 << 16)) return false;
    if (rowsA * std::max(a.lda0, a.lda1) * 2 >= 0x7ff00000LL) return false;
    if ((long long)a.Npad * a.K * 2 >= 0x7ff00000LL) return false;
    return true;
}

template <bool CONV, bool UPS, int NT, int VAR = 0, bool LO = false>
static void launch_big_lo(const LaunchCtx& ctx, const GemmDev& p, dim3 grid)
{
    auto kern = gemm_big_kernel<CONV, UPS, NT, VAR, LO>;
    allow_dynamic_lds((const void*)kern, smem_bytes(NT));
    hipLaunchKernelGGL(kern, grid, dim3(512), smem_bytes(NT), ctx.stream, p);
}
template <bool CONV, bool UPS, int NT, int VAR = 0>
static void launch_big_t(const LaunchCtx& ctx, const GemmDev& p, dim3 grid)
{
    if (p.Rlo || p.Dlo) {
        if constexpr (VAR == 0 && NT == 5) launch_big_lo<CONV, UPS, NT, 0, true>(ctx, p, grid);   // split-fp16 operands only occur at UNet widths
        else SDE_REQUIRE(false, SDE_ERR_INVALID_ARGUMENT, "gemm_big: split-fp16 epilogue is built for the 256x320 tile only");
    } else launch_big_lo<CONV, UPS, NT, VAR, false>(ctx, p, grid);
}

void launch_gemm_big(const LaunchCtx& ctx, GemmDev& p, int batch, bool conv)
{
    const int bn = p.Npad % 320 == 0 ? 320 : 256;
    p.nbm = cdiv(p.M, BM);
    p.nbn = p.Npad / bn;
    dim3 grid(p.nbm * p.nbn, p.splitk, batch);
#ifdef SDE_DIAG
    static const int var = getenv("SDE_BIGVAR") ? atoi(getenv("SDE_BIGVAR")) : 0;
    if (conv && !p.ups && var && bn == 320) {
        switch (var) {
        case 1: launch_big_t<true, false, 5, 1>(ctx, p, grid); return;
        case 2: launch_big_t<true, false, 5, 2>(ctx, p, grid); return;
        case 3: launch_big_t<true, false, 5, 3>(ctx, p, grid); return;
        case 8: launch_big_t<true, false, 5, 8>(ctx, p, grid); return;
        case 16: launch_big_t<true, false, 5, 16>(ctx, p, grid); return;
        case 32: launch_big_t<true, false, 5, 32>(ctx, p, grid); return;
        case 48: launch_big_t<true, false, 5, 48>(ctx, p, grid); return;
        default: break;
        }
    }
#endif
    if (bn == 320) {
        if (conv && p.ups) launch_big_t<true, true, 5>(ctx, p, grid);
        else if (conv)     launch_big_t<true, false, 5>(ctx, p, grid);
        else               launch_big_t<false, false, 5>(ctx, p, grid);
    } else {
        if (conv && p.ups) launch_big_t<true, true, 4>(ctx, p, grid);
        else if (conv)     launch_big_t<true, false, 4>(ctx, p, grid);
        else               launch_big_t<false, false, 4>(ctx, p, grid);
    }
}
