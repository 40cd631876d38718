// attention.hip — flash-style softmax(Q K^T * scale) V for gfx950, fp16 in / fp32 softmax.
//
// Used for the UNet's self-attention (S = 4096/1024/256/64, head dim 40/80/160 or 64) and
// cross-attention (S_kv = 77/152/227).  No mask in the model; keys beyond S_kv are masked here.
//
// MFMA path: block = 4 waves, each wave owns 32 query rows (two 16-wide tiles), the block
// walks K/V in tiles of 64 keys staged in LDS.  The score product is issued "swapped",
//   S^T[key][q] = K[key][:] . Q[q][:]      (v_mfma_f32_16x16x32_f16, A = K fragment, B = Q fragment)
// so the softmax axis (keys) lies in each lane's registers + the 4 lane groups, and the exponentiated
// S^T accumulators are *already* the B operand of the second product
//   O^T[d][q] += V^T[d][key] * P^T[key][q]
// whose A operand V^T is read from the row-major V tile with ds_read_b64_tr_b16 (hardware transpose).
// Online softmax in fp32 with exp2; O^T is rescaled per q (= per lane), no cross-lane traffic for it.
// Roofline: MFMA-bound; algorithmic flops 4*Sq*Skv*d per (batch, head).
//
// Plain path: one thread per (batch, head, query) — test sizes, odd head dims, cross-check.
#include <type_traits>

#include "kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

struct AttnDev {
    const half_t* Q; const half_t* K; const half_t* V; half_t* O;
    int ldq, ldk, ldv, ldo;
    int NB, heads, Sq, Skv, d;
    float scale_log2e;
};

// One K/V tile of work for a wave: scores, online softmax, PV.  MASK = the tile holds keys >= Skv.
// Softmax runs in the log2 domain on raw scores: p = exp2(s*c - m*c), c = scale*log2(e) (one FMA per element).
// The running max is only raised when it grows by more than RESCALE_THR (in log2 units): P then stays <= 2^THR,
// which fp16 P and fp32 accumulators tolerate, and the O rescale pass is skipped for almost every tile.
#define ATTN_RESCALE_THR 8.0f

// ONES: the head dim leaves a spare column in the last 16-wide V tile (d < 16*DVT); it is filled with 1.0 so the PV product
// also accumulates the softmax denominator (row DVT*16-1 of O^T) and the VALU row sum disappears.
// FOLD: the head dim leaves spare columns in the score product (d < DQK).  Q is pre-multiplied by scale*log2(e) and two
// spare columns carry -(running max) as an fp16 hi+lo pair against 1.0 in K, so the MFMA delivers s*c - m*c directly and the
// per-score VALU work is max + exp2 + cvt (the kernel is VALU-issue-bound at d = 40).
// LDS layouts, chosen against the gfx950 bank rules (MI355X_MICROARCH.md, LDS; PMC before: 42 % of LDS cycles were conflicts):
//  K, DQK = 64: unpadded 128-B rows with the 16-B chunk index XORed by (row & 7) -> ds_read_b128 conflict-free; other DQK keep
//  a one-chunk pad.  V (head dim <= 80): row stride = 32 B x odd -> the two 32-lane halves of ds_read_b64_tr_b16 each cover
//  all 64 banks once.
// DMA (head dims 33..64: SD1.x d=40, SD2.x d=64): K/V tiles go global -> LDS directly (buffer_load ... lds) into a 3-deep ring,
//  two tiles ahead, retired with a counted vmcnt; no staging registers, no ds_write.  Pad chunks (zeros, the FOLD / ONES
//  constants) are written once per ring slot and never touched again: the DMA lanes that would cover them are masked off.
// RING: depth of the LDS-DMA ring (DMA variants only).  3 = loads two tiles ahead, 2 blocks per CU by the 160-register budget the
// kernel then takes; 2 = one tile ahead, 28.7 KB per block and a 128-register budget: FOUR blocks per CU.
template <int DQK, int DVT, int RING = 3>
struct AttnLayout {
    static constexpr bool KSWZ = DQK == 64;
    static constexpr int KROW = KSWZ ? DQK : DQK + 8;                     // K-tile row (elements)
    static constexpr int VROW = DVT > 5 ? DVT * 16 + 8 : ((DVT & 1) ? DVT * 16 : DVT * 16 + 16);     // V-tile row (elements); wide heads: measured no gain
    static constexpr bool PREFETCH = DVT <= 4;   // wide heads: staging registers are not affordable, those levels are small
    static constexpr bool DMA = DQK == 64 || (DQK == 96 && DVT == 5);   // d = 33..64, and d = 80 (SD v1.x 32x32 level; round 3)
    static constexpr int NBUF = DMA ? RING : (PREFETCH ? 2 : 1);  // the register-staged prefetching variants double-buffer: one barrier per tile
    static constexpr int SMEM_BYTES = NBUF * 64 * (KROW + VROW) * 2;
};

// PERSIST (DMA variants, key range <= RING tiles: cross-attention): the block loads ALL of its head's K / V once, then walks query
// blocks qb, qb + gridDim.x, .. with no block barrier in that loop -- the one-block-per-query-block form spent most of a 77-key launch
// on per-block set-up (pad constants, the DMA round trip, two barriers) around two tiles of work.
template <int DQK, int DVT, bool ONES, bool FOLD, int RING = 3, bool PERSIST = false>
__global__ __launch_bounds__(256, (RING == 2 ? 4 : (DVT <= 6 ? 2 : 1))) void attn_mfma_kernel(AttnDev p)
{
    using L = AttnLayout<DQK, DVT, RING>;
    constexpr bool KSWZ = L::KSWZ, PREFETCH = L::PREFETCH, DMA = L::DMA;
    constexpr int KROW = L::KROW, VROW = L::VROW, NBUF = L::NBUF;
    constexpr int KS = DQK / 32;         // k-steps of the score product
    // dynamic LDS through a generic pointer (as the GEMM kernels do): with a static __shared__ array the compiler orders every
    // ds_read behind ALL outstanding LDS-DMA (s_waitcnt vmcnt(0)), which would serialise the ring
    extern __shared__ __attribute__((aligned(16))) char attn_smem[];
    half_t* const sK_ = (half_t*)attn_smem;
    half_t* const sV_ = sK_ + NBUF * 64 * KROW;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // XCD-aware block order (blocks b, b+8, ... share an XCD and its L2): every XCD gets a contiguous run of the flat
    // (batch, head, q-block) index, so the q-blocks that re-read one head's K/V sit behind one L2 (PMC: 3x over-fetch before)
    int nb, head, qb;
    {
        const int nqb = gridDim.x, per_head = nqb, total = nqb * gridDim.y * gridDim.z;
        int bid = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
        const int q8 = total >> 3, r8 = total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        qb = bid % per_head;
        const int hb = bid / per_head;
        head = hb % gridDim.y; nb = hb / gridDim.y;
    }
    int q0 = qb * 128 + wid * 32;
    const int d = p.d;
    const float csl = p.scale_log2e;

    const half_t* Qb = p.Q + (long long)nb * p.Sq * p.ldq + head * d;
    const half_t* Kb = p.K + (long long)nb * p.Skv * p.ldk + head * d;
    const half_t* Vb = p.V + (long long)nb * p.Skv * p.ldv + head * d;
    half_t* Ob = p.O + (long long)nb * p.Sq * p.ldo + head * d;

    // Q fragments (B operand): lane holds Q[q = q0 + 16*qt + fr][8*(4*ks+fg) .. +8]
    half8 qf[2][KS];
    auto load_q = [&]() {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int q = q0 + qt * 16 + fr, c = (ks * 4 + fg) * 8;
                half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (q < p.Sq && c < d) v = *(const half8*)(Qb + (long long)q * p.ldq + c);
                if (FOLD) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (half_t)((float)v[j] * csl);
                }
                qf[qt][ks] = v;
            }
    };
    load_q();
    // FOLD: column d (chunk ks_e, lane group fg_e) of Q holds -M_hi, column d+1 -M_lo (M = running max, log2 units); zero until the first tile is done
    const int ks_e = d >> 5, fg_e = (d >> 3) & 3;

    floatx4 o[DVT][2];
#pragma unroll
    for (int a = 0; a < DVT; ++a) { o[a][0] = floatx4{0, 0, 0, 0}; o[a][1] = floatx4{0, 0, 0, 0}; }
    float mrow[2] = {-INFINITY, -INFINITY}, lrow[2] = {0.f, 0.f};   // mrow in raw-score units

    // register-staged K/V tiles: the loads of tile t+1 are issued before tile t is computed.
    // Per-thread slots are fixed across tiles; rows past Skv are clamped to the last key (masked later), pad columns read zeros.
    constexpr int KCH = DQK / 8, VCH = DVT * 2;
    constexpr int KN = (64 * KCH + 255) / 256, VN = (64 * VCH + 255) / 256;
    half8 kreg[KN], vreg[VN];
    int krow[KN], vrow[VN], kcol[KN], vcol[VN];
    int klds[KN], vlds[VN];
#pragma unroll
    for (int i = 0; i < KN; ++i) {
        int idx = tid + i * 256;
        int r = idx / KCH, c = (idx - r * KCH) * 8;
        krow[i] = r; kcol[i] = (idx < 64 * KCH && c < d) ? c : -1;
        klds[i] = idx < 64 * KCH ? r * KROW + (KSWZ ? (((c >> 3) ^ (r & 7)) << 3) : c) : -1;
    }
#pragma unroll
    for (int i = 0; i < VN; ++i) {
        int idx = tid + i * 256;
        int r = idx / VCH, c = (idx - r * VCH) * 8;
        vrow[i] = r; vcol[i] = (idx < 64 * VCH && c < d) ? c : -1; vlds[i] = idx < 64 * VCH ? r * VROW + c : -1;
    }
    const int last_key = p.Skv - 1;
    // byte offsets of this thread's chunks inside a full tile (uniform tile base + 32-bit offset: no per-tile address math)
    unsigned koff[KN], voff[VN];
#pragma unroll
    for (int i = 0; i < KN; ++i) koff[i] = (unsigned)(krow[i] * p.ldk + max(kcol[i], 0)) * 2u;
#pragma unroll
    for (int i = 0; i < VN; ++i) voff[i] = (unsigned)(vrow[i] * p.ldv + max(vcol[i], 0)) * 2u;
    auto load_tile = [&](int kv0) {
        if (kv0 + 64 <= p.Skv) {   // block-uniform: every row of the tile exists
            const char* kt_ = (const char*)(Kb + (long long)kv0 * p.ldk);
            const char* vt_ = (const char*)(Vb + (long long)kv0 * p.ldv);
#pragma unroll
            for (int i = 0; i < KN; ++i) {
                half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kcol[i] >= 0) v = *(const half8*)(kt_ + koff[i]);
                kreg[i] = v;
            }
#pragma unroll
            for (int i = 0; i < VN; ++i) {
                half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (vcol[i] >= 0) v = *(const half8*)(vt_ + voff[i]);
                vreg[i] = v;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < KN; ++i) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kcol[i] >= 0) v = *(const half8*)(Kb + (long long)min(kv0 + krow[i], last_key) * p.ldk + kcol[i]);
            kreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (vcol[i] >= 0) v = *(const half8*)(Vb + (long long)min(kv0 + vrow[i], last_key) * p.ldv + vcol[i]);
            vreg[i] = v;
        }
    };
    auto store_tile = [&](int buf) {
        half_t* sK = sK_ + buf * 64 * KROW;
        half_t* sV = sV_ + buf * 64 * VROW;
#pragma unroll
        for (int i = 0; i < KN; ++i) if (kcol[i] >= 0) *(half8*)(sK + klds[i]) = kreg[i];
#pragma unroll
        for (int i = 0; i < VN; ++i) if (vcol[i] >= 0) *(half8*)(sV + vlds[i]) = vreg[i];
    };
    // pad columns (>= d) never change: written once.  ONES: the very last V column holds 1.0 (softmax denominator).
#pragma unroll
    for (int b = 0; b < (DMA ? 0 : NBUF); ++b) {
#pragma unroll
        for (int i = 0; i < KN; ++i)
            if (kcol[i] < 0 && klds[i] >= 0) {
                half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                if (FOLD && ((tid + i * 256) % KCH) * 8 == d) { z[0] = (half_t)1.0f; z[1] = (half_t)1.0f; }   // the two max-carrying columns
                *(half8*)(sK_ + b * 64 * KROW + klds[i]) = z;
            }
#pragma unroll
        for (int i = 0; i < VN; ++i)
            if (vcol[i] < 0 && vlds[i] >= 0) {
                half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ONES && vlds[i] - vrow[i] * VROW == DVT * 16 - 8) z[7] = (half_t)1.0f;
                *(half8*)(sV_ + b * 64 * VROW + vlds[i]) = z;
            }
    }

    // State of the tile in flight.  It lives outside the lambda because the LDS-DMA loop runs a tile in two steps when a running
    // maximum has to be raised (mode 1 = scores + the common-path finish, mode 2 = the rescaling finish, see that loop).
    floatx4 s[4][2];
    fp16x4 vlo[2][DVT], vhi[2][DVT];
    float mxq[2];
    bool trigq[2];
    // mode 0: the whole tile.  mode 1: scores; if no query of the wave raises its maximum, finish the tile and return false, else
    // return true with the tile unfinished.  mode 2: finish such a tile (rescale, then exp / PV).
    auto issue_v_reads = [&](const half_t* sV) {
        const int tq = fr >> 2, tp = fr & 3;
        const unsigned va = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)(sV + (fg * 4 + tq) * VROW + tp * 4);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int dt = 0; dt < DVT; ++dt) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[s2][dt]) : "v"(va), "n"((s2 * 32 * VROW + dt * 16) * 2));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[s2][dt]) : "v"(va), "n"((s2 * 32 * VROW + 16 * VROW + dt * 16) * 2));
            }
    };
    auto tile = [&](int kv0, int buf, auto mask_tag, auto mode_tag) -> bool {
        constexpr bool MASK = decltype(mask_tag)::value;
        constexpr int MODE = decltype(mode_tag)::value;
        const half_t* sK = sK_ + buf * 64 * KROW;
        const half_t* sV = sV_ + buf * 64 * VROW;
        const int tq = fr >> 2, tp = fr & 3;
        bool any_trig = false;
        if constexpr (MODE != 2) {
        // DMA path: the transposed V reads go out FIRST, ahead of the K-fragment reads.  LDS returns in order, so the waits the
        // compiler places for its own K reads (it does not know about the asm reads in front of them) also cover V, and the V
        // latency hides behind the score MFMAs instead of sitting between the running-max phase and the exponentials.
        if constexpr (DMA) issue_v_reads(sV);
        // scores S^T[key][q]: s[kt][qt][r] <-> key = kv0 + 16*kt + 4*fg + r, q = q0 + 16*qt + fr
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { s[kt][0] = floatx4{0, 0, 0, 0}; s[kt][1] = floatx4{0, 0, 0, 0}; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                half8 kf = *(const half8*)(sK + (kt * 16 + fr) * KROW + (KSWZ ? (((ks * 4 + fg) ^ (fr & 7)) << 3) : (ks * 4 + fg) * 8));
                s[kt][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[0][ks], s[kt][0], 0, 0, 0);
                s[kt][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[1][ks], s[kt][1], 0, 0, 0);
            }
        }
        // phase 1: mask, lane-local maxima, then ONE block-uniform branch: the common path (no query of this wave raises its
        // running max) goes straight to phase 2; the rare path rescales and then runs its own copy of phase 2.  With the branch
        // rejoining before phase 2 the compiler kept O and the Q fragments in two register sets and copied them across twice
        // per tile (40 v_mov_b64); with the two paths meeting only at the loop's back edge each keeps them where they are.
        bool trig = false;
        trigq[0] = trigq[1] = false;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (MASK) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (kv0 + kt * 16 + fg * 4 + r >= p.Skv) s[kt][qt][r] = -INFINITY;
            }
            // lane-local maximum of this lane's 16 scores of q-tile qt.  The query's other keys sit in the three lanes fr+16k;
            // the cross-lane reduction (two LDS round trips) is only paid when some lane sees the deferred-rescale trigger.
            float mx = s[0][qt][0];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][qt][r]);
            mxq[qt] = mx;
            // deferred rescale: raise the running max only when it grows by more than THR (log2 units)
            if (FOLD) trigq[qt] = mx > ATTN_RESCALE_THR || mrow[qt] == -INFINITY;   // scores are already s*c - M_old: mx is the growth
            else trigq[qt] = mx * csl > mrow[qt] * csl + ATTN_RESCALE_THR || mrow[qt] == -INFINITY;
            trig = trig || trigq[qt];
        }
        if constexpr (DMA) any_trig = __any(trig);
        }   // MODE != 2
        auto rescale = [&](int qt) {
            float mx = mxq[qt];
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            if (FOLD) {
                const bool first = mrow[qt] == -INFINITY;
                const bool grow = mx > ATTN_RESCALE_THR || first;
                const float delta = grow ? mx : 0.f;
                const float alpha = first ? 0.f : __builtin_amdgcn_exp2f(-delta);
                mrow[qt] = (first ? 0.f : mrow[qt]) + delta;
                lrow[qt] *= alpha;
            #pragma unroll
                for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dt][qt][r] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[kt][qt][r] -= delta;
                const half_t hi = (half_t)(-mrow[qt]);
                const half_t lo = (half_t)(-mrow[qt] - (float)hi);
                if (fg == fg_e) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        if (ks == ks_e) { qf[qt][ks][0] = hi; qf[qt][ks][1] = lo; }
                }
            } else {
                const bool grow = mx * csl > mrow[qt] * csl + ATTN_RESCALE_THR || mrow[qt] == -INFINITY;
                float mnew = grow ? mx : mrow[qt];
                float alpha = __builtin_amdgcn_exp2f((mrow[qt] - mnew) * csl);   // exp2(-inf) = 0 on the first tile
                if (mrow[qt] == mnew) alpha = 1.0f;
                mrow[qt] = mnew;
                lrow[qt] *= alpha;
            #pragma unroll
                for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[dt][qt][r] *= alpha;
            }
        };
        auto phase2 = [&]() {
        // phase 2 (straight line): V^T fragments, then exp / PV per q-tile so that the second tile's exponentials can
        // be scheduled between the first tile's PV MFMAs.
        // A operand via transposed LDS reads: lane 4q'+p' of a 16-lane group addresses row (key) q', columns 4p'..4p'+3
        // and receives column (lane&15).  k-slot (fg, j): j<4 -> key tile 2*s2, reg j ; j>=4 -> key tile 2*s2+1, reg j-4.
        auto read_v = [&](int s2, int dt) {
            const half_t* a0 = sV + (s2 * 32 + fg * 4 + tq) * VROW + dt * 16 + tp * 4;
            fp16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)a0);
            fp16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(a0 + 16 * VROW));
            half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
            half8 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = l4[r]; v[4 + r] = h4[r]; }
            return v;
        };
        constexpr bool HOLD_V = DVT <= 4 || DMA;   // keep the V^T fragments in registers for both q-tiles when they fit (the DMA variants read them from inline asm: always held)
        half8 vf[HOLD_V ? 2 : 1][HOLD_V ? DVT : 1];
        if constexpr (DMA) {
            // The transposed reads are issued from inline asm: the compiler orders every ds_read it can see behind ALL outstanding
            // LDS-DMA (s_waitcnt vmcnt(0)), which would drain the ring's prefetch.  Nothing else touches LDS between here and the
            // explicit lgkmcnt(0) below, so the compiler's own LDS bookkeeping is not disturbed.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // long since satisfied: the K-fragment waits were behind these reads
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt) {
                    asm volatile("" : "+v"(vlo[s2][dt]), "+v"(vhi[s2][dt]));   // values are defined only after the wait above
                    half4 l4 = __builtin_bit_cast(half4, vlo[s2][dt]), h4 = __builtin_bit_cast(half4, vhi[s2][dt]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { vf[HOLD_V ? s2 : 0][HOLD_V ? dt : 0][r] = l4[r]; vf[HOLD_V ? s2 : 0][HOLD_V ? dt : 0][4 + r] = h4[r]; }
                }
        } else if (HOLD_V) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt) vf[HOLD_V ? s2 : 0][HOLD_V ? dt : 0] = read_v(s2, dt);
        }
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            const float mc = mrow[qt] * csl;
            float sum = 0.f;
            half8 pf[2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e = FOLD ? __builtin_amdgcn_exp2f(s[kt][qt][r]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][qt][r], csl, -mc));
                    if (!ONES) sum += e;
                    pf[kt >> 1][(kt & 1) * 4 + r] = (half_t)e;
                }
            if (!ONES) lrow[qt] += sum;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt) {
                    half8 v = HOLD_V ? vf[HOLD_V ? s2 : 0][HOLD_V ? dt : 0] : read_v(s2, dt);
                    o[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(v, pf[s2], o[dt][qt], 0, 0, 0);
                }
        }
        };
        if constexpr (MODE == 0 && !DMA) {
            // register-staged variants (d <= 32, d >= 80): one short branch per q-tile that rejoins before phase 2; these kernels
            // sit at the VGPR limit and a second copy of phase 2 costs them more than the register copies at the join
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
                if (__any(trigq[qt])) rescale(qt);
            phase2();
            return false;
        } else if constexpr (MODE == 0) {
            if (any_trig) { rescale(0); rescale(1); phase2(); }
            else phase2();
            return false;
        } else if constexpr (MODE == 1) {
            if (any_trig) {
                // the transposed V reads issued from inline asm above are still in flight and the compiler does not know it: were the
                // fragments carried out of the loop as they are, it would be free to copy registers the LDS has not written yet
                // (so they are retired here, and the finishing call reads the fragments again: nothing of them is live across the loop exit)
                if constexpr (DMA) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                return true;
            }
            phase2();
            return false;
        } else {
            if constexpr (DMA) issue_v_reads(sV);
            rescale(0); rescale(1); phase2();
            return false;
        }
    };
    auto finish = [&]() {
    // finish: o[dt][qt][r] <-> d index 16*dt + 4*fg + r, q = q0 + 16*qt + fr
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        float l;
        if (ONES) l = __shfl(o[DVT - 1][qt][3], 48 + fr);   // row 16*DVT-1 of O^T lives in lane group 3, register 3
        else {
            l = lrow[qt];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
        }
        float inv = 1.0f / l;
        int q = q0 + qt * 16 + fr;
        if (q >= p.Sq) continue;
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt) {
            int c = dt * 16 + fg * 4;
            if (c < d) {
                half4 h = {(half_t)(o[dt][qt][0] * inv), (half_t)(o[dt][qt][1] * inv), (half_t)(o[dt][qt][2] * inv),
                           (half_t)(o[dt][qt][3] * inv)};
                *(half4*)(Ob + (long long)q * p.ldo + c) = h;
            }
        }
    }
    };
    using M0 = std::integral_constant<int, 0>;

    if constexpr (DMA) {
        constexpr int CPR = VROW / 8, VPW = 16 * CPR, NVI = (VPW + 63) / 64;            // V chunks per row / per wave; DMA instructions per wave
        constexpr int KCPR = KROW / 8, KPW = 16 * KCPR, NKI = (KPW + 63) / 64;          // K likewise: wave w fills rows 16 w .. 16 w + 15 (KROW = 64: two full pieces)
        const int wu = __builtin_amdgcn_readfirstlane(wid);
        const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, 0x80000000u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, 0x80000000u, 0x00020000);
        // lane -> (row, chunk) of the LDS positions it fills; the chunk's source column follows the K swizzle
        int k_r[NKI], k_cb[NKI], v_r[NVI], v_cb[NVI];
        unsigned k_off[NKI], v_off[NVI];
        bool k_act[NKI], v_act[NVI];
#pragma unroll
        for (int j = 0; j < NKI; ++j) {
            const int idw = j * 64 + lane, id = wu * KPW + idw, r = id / KCPR, cp = id - r * KCPR, c = KSWZ ? (cp ^ (r & 7)) : cp;
            k_r[j] = r; k_cb[j] = c * 16; k_act[j] = idw < KPW && c * 8 < d;
            k_off[j] = (unsigned)(r * p.ldk * 2 + c * 16);
        }
#pragma unroll
        for (int j = 0; j < NVI; ++j) {
            const int idw = j * 64 + lane, id = wu * VPW + idw, r = id / CPR, c = id - r * CPR;
            v_r[j] = r; v_cb[j] = c * 16; v_act[j] = idw < VPW && c * 8 < d;
            v_off[j] = (unsigned)(r * p.ldv * 2 + c * 16);
        }
        auto issue_tile = [&](int kv0, int slot) {
            char* kd = (char*)(sK_ + slot * 64 * KROW) + wu * KPW * 16;
            char* vd = (char*)(sV_ + slot * 64 * VROW) + wu * VPW * 16;
            const bool full = kv0 + 64 <= p.Skv;     // block-uniform
            const int ks_ = full ? kv0 * p.ldk * 2 : 0, vs_ = full ? kv0 * p.ldv * 2 : 0;
#pragma unroll
            for (int j = 0; j < NKI; ++j) {
                const unsigned vo = full ? k_off[j] : (unsigned)(min(kv0 + k_r[j], last_key) * p.ldk * 2 + k_cb[j]);
                if (k_act[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rK, (__attribute__((address_space(3))) void*)(kd + j * 1024), 16, vo, ks_, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NVI; ++j) {
                const unsigned vo = full ? v_off[j] : (unsigned)(min(kv0 + v_r[j], last_key) * p.ldv * 2 + v_cb[j]);
                if (v_act[j]) __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (__attribute__((address_space(3))) void*)(vd + j * 1024), 16, vo, vs_, 0, 0);
            }
        };
        // pad chunks of all three slots, once
#pragma unroll
        for (int b = 0; b < NBUF; ++b) {
#pragma unroll
            for (int i = 0; i < (64 * KCPR + 255) / 256; ++i) {
                const int id = tid + i * 256, r = id / KCPR, cp = id - r * KCPR, c = KSWZ ? (cp ^ (r & 7)) : cp;
                if (id < 64 * KCPR && c * 8 >= d) {
                    half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (FOLD && c * 8 == d) { z[0] = (half_t)1.0f; z[1] = (half_t)1.0f; }
                    *(half8*)(sK_ + b * 64 * KROW + id * 8) = z;
                }
            }
#pragma unroll
            for (int i = 0; i < (64 * CPR + 255) / 256; ++i) {
                const int id = tid + i * 256, r = id / CPR, c = id - r * CPR;
                if (id < 64 * CPR && c * 8 >= d) {
                    half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ONES && c == DVT * 2 - 1) z[7] = (half_t)1.0f;
                    *(half8*)(sV_ + b * 64 * VROW + id * 8) = z;
                }
            }
        }
        __syncthreads();
        const int ntile = (p.Skv + 63) >> 6;
        if constexpr (PERSIST) {
            // every tile of this head into its own slot (launcher: ntile <= RING), one wait, one barrier -- then no barrier any more
            for (int t = 0; t < ntile; ++t) issue_tile(t * 64, t);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const int nfullp = p.Skv >> 6, nqb = (p.Sq + 127) >> 7;
            for (int qbi = qb;;) {
                for (int t = 0; t < nfullp; ++t) tile(t * 64, t, std::false_type{}, M0{});
                if (nfullp < ntile) tile(nfullp * 64, nfullp, std::true_type{}, M0{});
                finish();
                qbi += (int)gridDim.x;
                if (qbi >= nqb) break;
                q0 = qbi * 128 + wid * 32;
                load_q();
#pragma unroll
                for (int a = 0; a < DVT; ++a) { o[a][0] = floatx4{0, 0, 0, 0}; o[a][1] = floatx4{0, 0, 0, 0}; }
                mrow[0] = mrow[1] = -INFINITY; lrow[0] = lrow[1] = 0.f;
            }
            return;
        }
        issue_tile(0, 0);
        if (RING == 3 && ntile > 1) issue_tile(64, 1);
        int slot = 0;
        const int nfull = p.Skv >> 6;
        // Full tiles.  The inner loop is the common path only: scores, exponentials, P.V, with O accumulated in place.  A tile whose
        // scores raise some query's running maximum leaves it unfinished (mode 1 returns true); the rescale and that tile's
        // exp / P.V then run out here and the inner loop is entered again.  Kept inside the loop, the rescale's writes to O and to
        // the Q fragments made the compiler hold every loop-carried register twice and copy them across on each tile.
        int t = 0;
        while (t < nfull) {
            bool unfinished = false;
            for (; t < nfull; ++t) {
                // this wave's pieces of tile t have landed (tile t+1's NKI+NVI pieces may stay in flight); the barrier then makes the
                // whole tile visible and tells everyone that slot (t+2)%3 == (t-1)%3 is no longer read
                if (RING == 3 && t + 1 < ntile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NKI + NVI) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (RING == 3) { if (t + 2 < ntile) issue_tile((t + 2) * 64, slot == 0 ? 2 : slot - 1); }
                else { if (t + 1 < ntile) issue_tile((t + 1) * 64, slot ^ 1); }      // the other slot was last read in tile t-1: the barrier above has passed it
                if (tile(t * 64, slot, std::false_type{}, std::integral_constant<int, 1>{})) { unfinished = true; break; }
                slot = RING == 3 ? (slot == 2 ? 0 : slot + 1) : (slot ^ 1);
            }
            if (!unfinished) break;
            (void)tile(t * 64, slot, std::false_type{}, std::integral_constant<int, 2>{});
            slot = RING == 3 ? (slot == 2 ? 0 : slot + 1) : (slot ^ 1);
            ++t;
        }
        if (nfull < ntile) {   // the ragged last tile, peeled: with both tile bodies inside the loop the compiler kept every loop-carried
                               // register (O, Q fragments) twice and copied them across on each iteration
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            tile(nfull * 64, slot, std::true_type{}, M0{});
        }
    } else
    if (PREFETCH) {
        // tile t is computed from LDS buffer t&1 while tile t+1 moves registers -> the other buffer and tile t+2 is in flight
        load_tile(0);
        store_tile(0);
        if (64 < p.Skv) load_tile(64);
        __syncthreads();
        int buf = 0;
#ifdef SDE_DIAG
        unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_t0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
        unsigned st_sum[3] = {0, 0, 0};
#define STAMP(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_sum[i] += (unsigned)(t_ - st_prev); st_prev = t_; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
        for (int kv0 = 0; kv0 < p.Skv; kv0 += 64, buf ^= 1) {
            if (kv0 + 64 < p.Skv) store_tile(buf ^ 1);     // last read in the previous iteration, which ended with a barrier
            if (kv0 + 128 < p.Skv) load_tile(kv0 + 128);
            STAMP(0);
            if (kv0 + 64 > p.Skv) tile(kv0, buf, std::true_type{}, M0{});   // block-uniform branch
            else tile(kv0, buf, std::false_type{}, M0{});
            STAMP(1);
            __syncthreads();
            STAMP(2);
        }
#ifdef SDE_DIAG
        if (blockIdx.x == 9 && blockIdx.y == 1 && blockIdx.z == 1 && lane == 0 && p.Skv >= 1024) {
            const unsigned tot = (unsigned)(__builtin_amdgcn_s_memtime() - st_t0), rt = (unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
            const int nt = p.Skv / 64;
            printf("[attn stamps] wave %d: per tile: store+load %u | tile %u | barrier wait %u ; loop %u cycles / %u x10ns = %.2f GHz\n", wid,
                   st_sum[0] / nt, st_sum[1] / nt, st_sum[2] / nt, tot, rt, tot / (rt * 10.0));
        }
#endif
#undef STAMP
    } else {
        for (int kv0 = 0; kv0 < p.Skv; kv0 += 64) {
            load_tile(kv0);
            __syncthreads();  // everyone done reading the previous tile
            store_tile(0);
            __syncthreads();
            if (kv0 + 64 > p.Skv) tile(kv0, 0, std::true_type{}, M0{});   // block-uniform branch
            else tile(kv0, 0, std::false_type{}, M0{});
        }
    }
    finish();
}

// ------------------------------------------------------------------------------------------ plain kernel
__global__ void attn_naive_kernel(AttnDev p)
{
    long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long total = (long long)p.NB * p.heads * p.Sq;
    if (gid >= total) return;
    int q = (int)(gid % p.Sq);
    int head = (int)((gid / p.Sq) % p.heads);
    int nb = (int)(gid / ((long long)p.Sq * p.heads));
    const int d = p.d;
    const half_t* Qr = p.Q + ((long long)nb * p.Sq + q) * p.ldq + head * d;
    const half_t* Kb = p.K + (long long)nb * p.Skv * p.ldk + head * d;
    const half_t* Vb = p.V + (long long)nb * p.Skv * p.ldv + head * d;
    half_t* Or = p.O + ((long long)nb * p.Sq + q) * p.ldo + head * d;
    float mx = -INFINITY;
    for (int k = 0; k < p.Skv; ++k) {
        float s = 0.f;
        for (int c = 0; c < d; ++c) s += (float)Qr[c] * (float)Kb[(long long)k * p.ldk + c];
        mx = fmaxf(mx, s * p.scale_log2e);
    }
    float l = 0.f;
    for (int k = 0; k < p.Skv; ++k) {
        float s = 0.f;
        for (int c = 0; c < d; ++c) s += (float)Qr[c] * (float)Kb[(long long)k * p.ldk + c];
        l += exp2f(s * p.scale_log2e - mx);
    }
    const float inv = 1.0f / l;
    for (int c0 = 0; c0 < d; c0 += 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < p.Skv; ++k) {
            float s = 0.f;
            for (int c = 0; c < d; ++c) s += (float)Qr[c] * (float)Kb[(long long)k * p.ldk + c];
            float e = exp2f(s * p.scale_log2e - mx);
            for (int j = 0; j < 8 && c0 + j < d; ++j) acc[j] += e * (float)Vb[(long long)k * p.ldv + c0 + j];
        }
        for (int j = 0; j < 8 && c0 + j < d; ++j) Or[c0 + j] = (half_t)(acc[j] * inv);
    }
}

bool attention_mfma_supported(const AttnArgs& a)
{
    if (a.d % 8 || a.d > 160) return false;
    if (a.ldq % 8 || a.ldk % 8 || a.ldv % 8 || a.ldo % 4) return false;
    if ((reinterpret_cast<uintptr_t>(a.Q) | reinterpret_cast<uintptr_t>(a.K) | reinterpret_cast<uintptr_t>(a.V)) & 15) return false;
    if (reinterpret_cast<uintptr_t>(a.O) & 7) return false;
    return true;
}

template <int DQK, int DVT>
static void launch_attn_t(const LaunchCtx& ctx, const AttnDev& p)
{
    dim3 grid(cdiv(p.Sq, 128), p.heads, p.NB);
    if constexpr (DQK == 64 && DVT == 3) {
        // d = 40 (SD v1.x): the two-slot ring, four blocks per CU (tuning hook SDE_ATTN_RING=3: the three-slot ring, two to three per CU)
        static const int ring_env = getenv("SDE_ATTN_RING") ? atoi(getenv("SDE_ATTN_RING")) : 2;
        static const int fold_env2 = getenv("SDE_ATTN_FOLD") ? atoi(getenv("SDE_ATTN_FOLD")) : 1;
        // short key ranges (cross-attention: two tiles) keep the deeper ring: with one tile of look-ahead their few loads are all exposed (47.6 -> 54 us measured)
        // cross-attention (key range within the three ring slots): persistent blocks, two per CU, each walking query blocks of its head
        static const int persist_env = getenv("SDE_ATTN_PERSIST") ? atoi(getenv("SDE_ATTN_PERSIST")) : 1;   // tuning hook
        if (persist_env && fold_env2 && p.d < DQK && p.d < DVT * 16 && p.Skv <= 192 && p.Sq >= 1024) {
            const int nqb = cdiv(p.Sq, 128), pairs = p.heads * p.NB;
            const int G = std::max(1, std::min(nqb, cdiv(512, pairs)));      // two resident blocks per CU (213 registers)
            hipLaunchKernelGGL((attn_mfma_kernel<DQK, DVT, true, true, 3, true>), dim3(G, p.heads, p.NB), dim3(256), (AttnLayout<DQK, DVT, 3>::SMEM_BYTES),
                               ctx.stream, p);
            return;
        }
        if (ring_env == 2 && fold_env2 && p.d < DQK && p.d < DVT * 16 && p.Skv >= 512) {
            hipLaunchKernelGGL((attn_mfma_kernel<DQK, DVT, true, true, 2>), grid, dim3(256), (AttnLayout<DQK, DVT, 2>::SMEM_BYTES), ctx.stream, p);
            return;
        }
    }
    constexpr int smem = AttnLayout<DQK, DVT>::SMEM_BYTES;
    static const int fold_env = getenv("SDE_ATTN_FOLD") ? atoi(getenv("SDE_ATTN_FOLD")) : 1;   // tuning hook
    const bool fold = fold_env && p.d < DQK && DVT <= 4;    // needs two spare score-product columns; wider heads: measured slower on the staged path (VGPRs), no gain on the d = 80 DMA path (78.3 against 77.6 us)
    const bool ones = p.d < DVT * 16;
    auto go = [&](auto kern) {
        if (smem > 64 * 1024) allow_dynamic_lds((const void*)kern, smem);
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, ctx.stream, p);
    };
    if (fold) {
        if (ones) go(attn_mfma_kernel<DQK, DVT, true, true>); else go(attn_mfma_kernel<DQK, DVT, false, true>);
    } else {
        if (ones) go(attn_mfma_kernel<DQK, DVT, true, false>); else go(attn_mfma_kernel<DQK, DVT, false, false>);
    }
}

void launch_attention(const LaunchCtx& ctx, const AttnArgs& a)
{
    SDE_REQUIRE(a.NB > 0 && a.heads > 0 && a.Sq > 0 && a.Skv > 0 && a.d > 0, SDE_ERR_INVALID_ARGUMENT, "attention: empty problem");
    if (ctx.f32) { f32_launch_attention(ctx, a); return; }
    AttnDev p{a.Q, a.K, a.V, a.O, a.ldq, a.ldk, a.ldv, a.ldo, a.NB, a.heads, a.Sq, a.Skv, a.d,
              a.scale * 1.44269504088896340736f};
    const double flops = 4.0 * a.NB * a.heads * (double)a.Sq * a.Skv * a.d;
    const double bytes = 2.0 * a.NB * a.heads * ((double)a.Sq * a.d * 2 + (double)a.Skv * a.d * 2);
    if (!ctx.force_naive && attention_mfma_supported(a)) {
        const int d = a.d;
        // one profiler class per kernel instance (self- and cross-attention of one head dim share it), named so that it can be
        // matched against the kernel symbol rocprofv3 reports: attn_mfma<DQK,DVT> <-> attn_mfma_kernel<DQK, DVT, ONES, FOLD>
        static thread_local char cls[32];
        {
            const int dvt = d <= 16 ? 1 : (d <= 32 ? 2 : (d <= 96 ? (d + 15) / 16 : (d <= 128 ? 8 : 10)));
            const int dqk = d <= 32 ? 32 : (d <= 64 ? 64 : (d <= 96 ? 96 : (d <= 128 ? 128 : 160)));
            snprintf(cls, sizeof(cls), "attn_mfma<%d,%d>", dqk, dvt);
        }
        ProfScope ps(ctx, cls, flops, bytes);
        if (d <= 16) launch_attn_t<32, 1>(ctx, p);
        else if (d <= 32) launch_attn_t<32, 2>(ctx, p);
        else if (d <= 48) launch_attn_t<64, 3>(ctx, p);
        else if (d <= 64) launch_attn_t<64, 4>(ctx, p);
        else if (d <= 80) launch_attn_t<96, 5>(ctx, p);
        else if (d <= 96) launch_attn_t<96, 6>(ctx, p);
        else if (d <= 128) launch_attn_t<128, 8>(ctx, p);
        else launch_attn_t<160, 10>(ctx, p);
    } else {
        ProfScope ps(ctx, "attn_plain", flops, bytes);
        long long total = (long long)a.NB * a.heads * a.Sq;
        hipLaunchKernelGGL(attn_naive_kernel, dim3((unsigned)cdiv64(total, 128)), dim3(128), 0, ctx.stream, p);
    }
    HIP_CHECK(hipGetLastError());
}
