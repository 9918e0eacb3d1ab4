// Fused attention softmax(Q K^T * scale) V for gfx950, head_dim 64 / 128, bf16 in, fp32 softmax.
// Replaces F.scaled_dot_product_attention under diffusers' AttnProcessor2_0 / FluxAttnProcessor2_0
// (UNet self/cross attention, Flux joint attention).
//
// CDNA4 design notes
//  * one workgroup = 4 waves (one per SIMD) = 128 queries; a wave owns 32 queries for the whole
//    key sweep, K/V tiles of 64 keys are shared by the 4 waves through LDS (register-staged double
//    buffering, one barrier per tile)
//  * scores are computed TRANSPOSED, S^T = K Q^T, with v_mfma_f32_32x32x16_bf16: the C layout then
//    puts one query per lane (col = lane & 31) and its keys in registers, so the online-softmax
//    row statistics are register-local plus one exchange with lane ^ 32
//  * O is accumulated transposed too (O^T = V^T P^T), so the per-query rescale factor is
//    lane-local; the P^T operand is the score registers converted in place: the MFMA contraction
//    index is a PERMUTATION of the key index (slot j of half h <-> key (j&3) + 8*(j>>2) + 4h),
//    matched on the V side by reading two 8-byte pieces of the transposed V tile - no cross-lane
//    shuffles, no LDS round trip for P
//  * V arrives transposed from the projection GEMM's epilogue (OMGSR_LAYOUT_T), so the key index
//    is contiguous and every operand fragment is a 16-byte (K) or 2 x 8-byte (V^T) LDS read
//  * LDS pitches: K rows 2*D+16 B (odd number of 16-B slots -> conflict-free ds_read_b128),
//    V^T rows 136 B (34 banks -> conflict-free ds_read_b64)
//  * DMA = true (Lk % 64 == 0: every self-attention of the UNet and the Flux joint attention): the K / V^T tiles go
//    global -> LDS with `global_load_lds_dwordx4` (1 KiB per wave instruction, no staging registers, no ds_write: the
//    register-staged form spends ~400 LDS cycles per tile on the wide stores, next to ~500 of fragment reads, against 1024
//    MFMA cycles). A DMA piece lands linearly (lane i -> 16 B at 16 i), so rows are unpadded and the 16-byte slot of a
//    chunk is XOR-swizzled with the row instead (each lane FETCHES the chunk that belongs at its slot): conflict-free
//    ds_read_b128 for K and V^T alike. The V^T fragment becomes ONE 16-byte read because the K rows of a 32-key block are
//    read through the permutation that swaps bits 2 and 3 of the row index: C row (r&3) + 8(r>>2) + 4h of S^T then holds
//    key 16(r>>3) + 8h + (r&7), i.e. the P^T operand's contraction slots are 8 CONSECUTIVE keys.
#include "common.hip.h"
#include <type_traits>
#include "../../include/omgsr_hip.h"
#include "timing.hip.h"
#include <stdlib.h>

namespace {

OMGSR_DEVINL void glds16_sv(const unsigned voff, const void* sbase, const unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// SPLIT (round 6, D = 64; omgsr_attn_args.q_lo_off / k_lo_off / p_split): q, k and the probabilities as two-term splits of 16-bit values - the
// range-fallback tier's bf16 operands carry 8-bit mantissas, and the UNet's 32 attention calls per tile with single bf16 q / k / P cost that tier
// the north-star tolerance (q 28, k 29, P 26 of its 132 units of squared error). S^T = K_hi Q_hi^T + K_lo Q_hi^T + K_hi Q_lo^T (the
// K_lo tile rides through LDS behind K_hi: same pieces, same swizzle, same fragment offsets), O^T += V^T P_hi^T + V^T P_lo^T.
template <typename T, int D, bool DMA, bool SPLIT>
__global__ __launch_bounds__(256, 2) void attn_kernel(const omgsr_attn_args p, const int ntiles, const float defer, const int qtiles, const int xcd_order) {
    constexpr int KP = DMA ? 2 * D : 2 * D + 16;
    constexpr int VP = DMA ? 128 : 136;
    constexpr int K_BYTES = 64 * KP, V_BYTES = D * VP;
    constexpr int KT_BYTES = SPLIT ? 2 * K_BYTES : K_BYTES;          // [K_hi | K_lo] of a tile
    constexpr int VT_BYTES = SPLIT ? 2 * V_BYTES : V_BYTES;          // [V^T_hi | V^T_lo] (the lo tile is only filled / read when p.vt_lo_off > 0)
    constexpr int STAGE = KT_BYTES + VT_BYTES;
    constexpr int CPR = D / 8;               // 16-byte chunks per K row
    constexpr int NKC = 64 * CPR / 256;      // K chunks per thread
    constexpr int NVC = D * 8 / 256;         // V^T chunks per thread
    constexpr int NKS = D / 16, NDB = D / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    // 1-D grid of qtiles x H x B workgroups. XCD-aware order (round 6): the hardware deals consecutive block ids round-robin over the 8 XCDs, so
    // with the natural order every XCD's L2 fetched every head's K / V^T (FLUX: 8 x 453 MB per launch = the 4.9x of profiles/r05_traffic_f1024_*);
    // xcd_remap gives each XCD a contiguous range of (b, h, q-tile) ids: the 36 (FLUX) / 32 (UNet) query tiles of a head share ONE L2.
    const int tile = xcd_order ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int qt = tile % qtiles, bh = tile / qtiles;
    const int h = bh % p.H, b = bh / p.H;
    const int q0 = qt * 128 + wave * 32;

    const T* __restrict__ qp = (const T*)p.q + (int64_t)b * p.q_bstride + h * D;
    const T* __restrict__ kp = (const T*)p.k + (int64_t)b * p.k_bstride + h * D;
    const T* __restrict__ vp = (const T*)p.vt + (int64_t)b * p.vt_bstride + (int64_t)h * D * p.vt_ld;

    // Q^T operand fragments live in registers for the whole sweep
    x8_t<T> qf[NKS], qfl[SPLIT ? NKS : 1];
    {
        int qrow = q0 + l31; if (qrow > p.Lq - 1) qrow = p.Lq - 1;
        const T* qr = qp + (int64_t)qrow * p.q_ld + 8 * half;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) qf[ks] = *reinterpret_cast<const x8_t<T>*>(qr + 16 * ks);
        if constexpr (SPLIT) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) qfl[ks] = *reinterpret_cast<const x8_t<T>*>(qr + p.q_lo_off + 16 * ks);
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qfl[ks]));
        }
        // retire the loads HERE: hipcc otherwise places their counted vmcnt waits at the first use inside the key loop, where
        // they also wait for the LDS-DMA of the next tile (vmcnt(7) ... vmcnt(0) across the first eight MFMAs: no prefetch left)
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    // ---- DMA staging: wave w moves pieces w, w + 4, ... of the K tile and of the V^T tile (1 KiB each); a lane's source offset
    // is the same for all of its pieces (the swizzle term repeats every 4 pieces), the piece / tile offsets go into the SGPR base
    constexpr int CPRK = D / 8, RPPK = 64 / CPRK, RPBK = 16 / CPRK;      // K: chunks per row, rows per piece, rows per 256 B of banks
    constexpr int NPW = (64 * 2 * D / 1024) / 4;                         // pieces per wave and operand (D = 128: 4, 64: 2)
    unsigned kvoff = 0, vvoff = 0;
    if constexpr (DMA) {
        const int krow = RPPK * wave + lane / CPRK, kc = (lane % CPRK) ^ ((krow / RPBK) & (CPRK - 1));
        kvoff = (unsigned)((krow * p.k_ld + kc * 8) * 2);
        const int vrow = 8 * wave + (lane >> 3), vc = (lane & 7) ^ ((vrow >> 1) & 7);
        vvoff = (unsigned)((vrow * p.vt_ld + vc * 8) * 2);
    }
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;
    // piece i of the wave's NT * NPW per tile (NT = 2, or 4 with the K_lo and V^T_lo tiles): type i % NT = K (hi) / V^T / K_lo / V^T_lo, index i / NT
    constexpr int NT = SPLIT ? 4 : 2;
    const bool vsplit = SPLIT && p.vt_lo_off > 0;                    // (wave-uniform: a kernel argument)
    auto issue_piece = [&](const int kt, const int buf, const int i) {
        const int j = i / NT, ty = i % NT;
        const unsigned dst = lds_base + buf * STAGE + wave * 1024 + j * 4096;
        if (ty == 1 || ty == 3) {
            if (ty == 3 && !vsplit) return;
            const unsigned char* vb = reinterpret_cast<const unsigned char*>(vp + (ty == 3 ? p.vt_lo_off : 0) + (int64_t)kt * 64);
            glds16_sv(vvoff, vb + (int64_t)j * 32 * p.vt_ld * 2, __builtin_amdgcn_readfirstlane(dst + KT_BYTES + (ty == 3 ? V_BYTES : 0)));
        } else {
            const unsigned char* kb = reinterpret_cast<const unsigned char*>(kp + (ty == 2 ? p.k_lo_off : 0) + (int64_t)kt * 64 * p.k_ld);
            glds16_sv(kvoff, kb + (int64_t)j * (4 * RPPK) * p.k_ld * 2, __builtin_amdgcn_readfirstlane(dst + (ty == 2 ? K_BYTES : 0)));
        }
    };
    auto issue_tile = [&](const int kt, const int buf) {
#pragma unroll
        for (int i = 0; i < NT * NPW; ++i) issue_piece(kt, buf, i);
    };

    u32x4_t kreg[DMA ? 1 : NKC], vreg[DMA ? 1 : NVC], kreg_lo[(!DMA && SPLIT) ? NKC : 1], vreg_lo[(!DMA && SPLIT) ? NVC : 1];
    auto load_tile = [&](int kt) {
        const int key_base = kt * 64;
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = t + 256 * i;
            const int row = c / CPR, kc = c % CPR;
            const int key = key_base + row;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (key < p.Lk) v = *reinterpret_cast<const u32x4_t*>(kp + (int64_t)key * p.k_ld + kc * 8);
            kreg[i] = v;
            if constexpr (SPLIT) {
                u32x4_t vl = {0u, 0u, 0u, 0u};
                if (key < p.Lk) vl = *reinterpret_cast<const u32x4_t*>(kp + p.k_lo_off + (int64_t)key * p.k_ld + kc * 8);
                kreg_lo[i] = vl;
            }
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = t + 256 * i;
            const int drow = c >> 3, kc = c & 7;
            const int key0 = key_base + kc * 8;
            const int nvalid = p.Lk - key0;
            u32x4_t v = {0u, 0u, 0u, 0u}, vl = {0u, 0u, 0u, 0u};
            if (nvalid > 0) {
                v = *reinterpret_cast<const u32x4_t*>(vp + (int64_t)drow * p.vt_ld + key0);
                if constexpr (SPLIT) {
                    if (vsplit) vl = *reinterpret_cast<const u32x4_t*>(vp + p.vt_lo_off + (int64_t)drow * p.vt_ld + key0);
                }
                if (nvalid < 8) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        if (2 * w >= nvalid) { v[w] = 0u; vl[w] = 0u; }
                        else if (2 * w + 1 >= nvalid) { v[w] &= 0xffffu; vl[w] &= 0xffffu; }
                    }
                }
            }
            vreg[i] = v;
            if constexpr (SPLIT) vreg_lo[i] = vl;
        }
    };
    auto write_tile = [&](int buf) {
        unsigned char* Ks = lds + buf * STAGE;
        unsigned char* Vs = Ks + KT_BYTES;
#pragma unroll
        for (int i = 0; i < NKC; ++i) {
            const int c = t + 256 * i;
            *reinterpret_cast<u32x4_t*>(Ks + (c / CPR) * KP + (c % CPR) * 16) = kreg[i];
            if constexpr (SPLIT) *reinterpret_cast<u32x4_t*>(Ks + K_BYTES + (c / CPR) * KP + (c % CPR) * 16) = kreg_lo[i];
        }
#pragma unroll
        for (int i = 0; i < NVC; ++i) {
            const int c = t + 256 * i;
            unsigned char* d = Vs + (c >> 3) * VP + (c & 7) * 16;     // 8-byte aligned only
            *reinterpret_cast<u32x2_t*>(d) = (u32x2_t){vreg[i][0], vreg[i][1]};
            *reinterpret_cast<u32x2_t*>(d + 8) = (u32x2_t){vreg[i][2], vreg[i][3]};
            if constexpr (SPLIT) {
                *reinterpret_cast<u32x2_t*>(d + V_BYTES) = (u32x2_t){vreg_lo[i][0], vreg_lo[i][1]};
                *reinterpret_cast<u32x2_t*>(d + V_BYTES + 8) = (u32x2_t){vreg_lo[i][2], vreg_lo[i][3]};
            }
        }
    };

    f32x16_t o[NDB];
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float sc = p.scale * 1.4426950408889634f;   // softmax in base 2

    // DMA: per-lane fragment offsets inside a stage. K: row 32 sb + pi(l31) (pi swaps bits 2 and 3), chunk 2 ks + half at
    // slot chunk ^ f(row); V^T: row 32 db + l31, chunk 4 sb + 2 u + half. The block terms (32 sb rows, 32 db rows) leave the
    // swizzle term unchanged and are immediates.
    unsigned koff[DMA ? NKS : 1], voff[DMA ? 4 : 1];
    if constexpr (DMA) {
        const int prow = (l31 & ~12) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
        const int fk = (prow / RPBK) & (CPRK - 1);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) koff[ks] = (unsigned)(prow * KP + (((2 * ks + half) ^ fk) << 4));
        const int fv = (l31 >> 1) & 7;
#pragma unroll
        for (int c = 0; c < 4; ++c) voff[c] = (unsigned)(l31 * VP + (((2 * c + half) ^ fv) << 4));
        issue_tile(0, 0);
    } else {
        load_tile(0);
        write_tile(0);
        __syncthreads();
    }

    for (int kt = 0; kt < ntiles; ++kt) {
        const int buf = kt & 1;
        if constexpr (DMA) {
            // tile kt has landed (this wave's pieces; the barrier extends that to every wave's) and every fragment read of
            // tile kt - 1 has returned, so its stage may be overwritten
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // issued here, in front of the QK MFMAs: one piece between every two MFMAs measured 3 % slower (901 vs 931 TFLOP/s)
            if (kt + 1 < ntiles) issue_tile(kt + 1, buf ^ 1);
        } else {
            if (kt + 1 < ntiles) load_tile(kt + 1);
        }
        const unsigned char* Ks = lds + buf * STAGE;
        const unsigned char* Vs = Ks + KT_BYTES;

        // S^T = K Q^T : two 32-key blocks
        f32x16_t s[2];
        x8_t<T> vf[DMA ? NDB : 1][4], vfl[(DMA && SPLIT) ? NDB : 1][(DMA && SPLIT) ? 4 : 1];
        if constexpr (DMA) {
            // every K fragment of the tile is requested before the first MFMA (left to itself hipcc reuses ONE fragment
            // register: ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma, 32 exposed LDS latencies per tile); sched_barriers pin the phases
            x8_t<T> kf[2][NKS], kfl[SPLIT ? 2 : 1][SPLIT ? NKS : 1];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) kf[sb][ks] = *reinterpret_cast<const x8_t<T>*>(Ks + 32 * sb * KP + koff[ks]);
            if constexpr (SPLIT) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) kfl[sb][ks] = *reinterpret_cast<const x8_t<T>*>(Ks + K_BYTES + 32 * sb * KP + koff[ks]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sb][r] = 0.0f;
            __builtin_amdgcn_s_setprio(1);
            if constexpr (SPLIT) {          // the two correction products first (small terms into the empty accumulator), then the main product
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) s[sb] = mfma32(kfl[sb][ks], qf[ks], s[sb]);
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) s[sb] = mfma32(kf[sb][ks], qfl[ks], s[sb]);
            }
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) s[sb] = mfma32(kf[sb][ks], qf[ks], s[sb]);
            __builtin_amdgcn_s_setprio(0);
            // ... and the V^T fragments before the softmax, whose ~1000 VALU cycles cover their latency
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int c = 0; c < 4; ++c) vf[db][c] = *reinterpret_cast<const x8_t<T>*>(Vs + 32 * db * VP + voff[c]);
            if constexpr (SPLIT) {
                if (vsplit) {
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
#pragma unroll
                        for (int c = 0; c < 4; ++c) vfl[db][c] = *reinterpret_cast<const x8_t<T>*>(Vs + V_BYTES + 32 * db * VP + voff[c]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[sb][r] = 0.0f;
                const unsigned char* kr = Ks + (32 * sb + l31) * KP + half * 16;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const x8_t<T> kl = *reinterpret_cast<const x8_t<T>*>(kr + K_BYTES + ks * 32);
                        s[sb] = mfma32(kl, qf[ks], s[sb]);
                    }
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        const x8_t<T> kf = *reinterpret_cast<const x8_t<T>*>(kr + ks * 32);
                        s[sb] = mfma32(kf, qfl[ks], s[sb]);
                    }
                }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const x8_t<T> kf = *reinterpret_cast<const x8_t<T>*>(kr + ks * 32);
                    s[sb] = mfma32(kf, qf[ks], s[sb]);
                }
            }
        }
        // online softmax (one query per lane pair), base 2. The raw scores stay unscaled: the row maximum is
        // taken on them (scale > 0), and p = exp2(fma(s, sc, -m*sc)) folds scale, max-subtract and the base change
        // into ONE fma + v_exp_f32 per score. The O / l rescale is skipped when no row of the wave saw its
        // maximum move (the common case after the first tiles).
        const bool tail = !DMA && (kt == ntiles - 1) && (p.Lk & 63);
        float mt = -INFINITY;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (tail && (kt * 64 + 32 * sb + cfrag_row(lane, r) >= p.Lk)) s[sb][r] = -INFINITY;
                mt = fmaxf(mt, s[sb][r]);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        // Deferred maximum (omgsr_set_attention_defer_max, fast tiers): the running maximum (and with it O, l) is only moved when some
        // row of the wave saw its maximum grow by more than 2^defer in the scaled base-2 domain; until then p = exp2((s - m_run) * sc)
        // <= 2^defer, inside the 16-bit operand range, and O / l is the same quotient. On smooth score distributions that is the
        // first tile only (with defer = 0, i.e. the exact running maximum, ~half of the tiles of a 4608-key sweep rescale the 64 O
        // registers): +5 % throughput; the dominant weight of a row is then a rounded exp2(delta) instead of an exact 1, which costs
        // ~1 % of the pipeline's error budget - the accurate tier keeps defer = 0.
        const float m_new = fmaxf(m_run, mt);
        if (__any((m_new - m_run) * sc > defer)) {                                // m_run = -inf on the first tile -> inf > defer
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sc);    // ... and alpha = 0
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
            m_run = m_new;
        }
        const float neg_m = -m_run * sc;
        float rs = 0.0f;
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(fmaf(s[sb][r], sc, neg_m));
                s[sb][r] = e;
                rs += e;
            }
        l_run += rs;

        // P^T operand: score registers converted in place (key permutation, see header)
        x8_t<T> pf[2][2], pfl[SPLIT ? 2 : 1][SPLIT ? 2 : 1];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[sb][u][j] = (T)s[sb][8 * u + j];
        const bool psplit = SPLIT && p.p_split;
        if constexpr (SPLIT) {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pfl[sb][u][j] = (T)(s[sb][8 * u + j] - (float)pf[sb][u][j]);
        }

        // O^T += V^T P^T
        if constexpr (DMA) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if constexpr (SPLIT) {
                if (psplit) {
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                            for (int u = 0; u < 2; ++u) o[db] = mfma32(vf[db][2 * sb + u], pfl[sb][u], o[db]);
                }
                if (vsplit) {
#pragma unroll
                    for (int db = 0; db < NDB; ++db)
#pragma unroll
                        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                            for (int u = 0; u < 2; ++u) o[db] = mfma32(vfl[db][2 * sb + u], pf[sb][u], o[db]);
                }
            }
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int u = 0; u < 2; ++u) o[db] = mfma32(vf[db][2 * sb + u], pf[sb][u], o[db]);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int db = 0; db < NDB; ++db) {
                const unsigned char* vr = Vs + (32 * db + l31) * VP + 8 * half;
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const unsigned char* a = vr + (32 * sb + 16 * u) * 2;
                        const u32x2_t lo = *reinterpret_cast<const u32x2_t*>(a);
                        const u32x2_t hi = *reinterpret_cast<const u32x2_t*>(a + 16);
                        const u32x4_t both = {lo[0], lo[1], hi[0], hi[1]};
                        if constexpr (SPLIT) {
                            if (psplit) o[db] = mfma32(*reinterpret_cast<const x8_t<T>*>(&both), pfl[sb][u], o[db]);
                            if (vsplit) {
                                const u32x2_t llo = *reinterpret_cast<const u32x2_t*>(a + V_BYTES);
                                const u32x2_t lhi = *reinterpret_cast<const u32x2_t*>(a + V_BYTES + 16);
                                const u32x4_t bl = {llo[0], llo[1], lhi[0], lhi[1]};
                                o[db] = mfma32(*reinterpret_cast<const x8_t<T>*>(&bl), pf[sb][u], o[db]);
                            }
                        }
                        o[db] = mfma32(*reinterpret_cast<const x8_t<T>*>(&both), pf[sb][u], o[db]);
                    }
            }
        }
        if constexpr (!DMA) {
            if (kt + 1 < ntiles) write_tile(buf ^ 1);
            __syncthreads();
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + l31;
    if (qrow < p.Lq) {
        T* op = (T*)p.o + (int64_t)b * p.o_bstride + (int64_t)qrow * p.o_ld + h * D + 4 * half;
        const int lo_off = p.o_lo_off;       // > 0: the low halves of the two-term split land that many columns later
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v0 = o[db][4 * g] * inv, v1 = o[db][4 * g + 1] * inv, v2 = o[db][4 * g + 2] * inv, v3 = o[db][4 * g + 3] * inv;
                u32x2_t w;
                w[0] = pack2<T>(v0, v1);
                w[1] = pack2<T>(v2, v3);
                *reinterpret_cast<u32x2_t*>(op + 32 * db + 8 * g) = w;
                if (p.o_mx) {
                    // OMGSR_EL_MX row (the consumer is an MX GEMM): [hi fp16 (2C B) | (v - hi) 2^11 fp8 (C B) | hi fp8 (C B)], C = H * D
                    // channels; o_ld counts 16-bit slots of the row (2C). Channel c = h * D + 4 half + 32 db + 8 g.
                    if constexpr (std::is_same<T, f16_t>::value) {
                        const x8_t<T> hv = __builtin_bit_cast(x8_t<T>, (u32x4_t){w[0], w[1], 0u, 0u});
                        const float h0 = (float)hv[0], h1 = (float)hv[1], h2 = (float)hv[2], h3 = (float)hv[3];
                        const float sc = (float)(1 << OMGSR_MX_LO_SHIFT);
                        unsigned char* row = reinterpret_cast<unsigned char*>((T*)p.o + (int64_t)b * p.o_bstride + (int64_t)qrow * p.o_ld);
                        const int Cc = p.H * D, c = h * D + 4 * half + 32 * db + 8 * g;
                        *reinterpret_cast<unsigned int*>(row + 2 * Cc + c) = pack4_fp8((v0 - h0) * sc, (v1 - h1) * sc, (v2 - h2) * sc, (v3 - h3) * sc);
                        *reinterpret_cast<unsigned int*>(row + 3 * Cc + c) = pack4_fp8(h0, h1, h2, h3);
                    }
                } else if (lo_off > 0) {
                    const x8_t<T> hv = __builtin_bit_cast(x8_t<T>, (u32x4_t){w[0], w[1], 0u, 0u});
                    u32x2_t l;
                    l[0] = pack2<T>(v0 - (float)hv[0], v1 - (float)hv[1]);
                    l[1] = pack2<T>(v2 - (float)hv[2], v3 - (float)hv[3]);
                    *reinterpret_cast<u32x2_t*>(op + lo_off + 32 * db + 8 * g) = l;
                }
            }
    }
}

float g_defer_max = 0.0f;

template <int D, bool DMA, bool SPLIT = false>
int launch_attn(const omgsr_attn_args& a, hipStream_t st) {
    constexpr int KT = SPLIT ? 2 : 1;
    constexpr int LDS = DMA ? 2 * KT * (64 * 2 * D + D * 128) : 2 * KT * (64 * (2 * D + 16) + D * 136);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<bf16_t, D, DMA, SPLIT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_kernel<f16_t, D, DMA, SPLIT>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int ntiles = (a.Lk + 63) / 64;
    const int qtiles = (a.Lq + 127) / 128;
    const int64_t blocks = (int64_t)qtiles * a.H * a.B;
    if (blocks > 0x7fffffffll) return OMGSR_E_SHAPE;
    static const char* xo = getenv("OMGSR_ATTN_XCD");            // A/B runs: "0" = natural block order (every XCD reads every head's K / V^T)
    const int xcd_order = !(xo && xo[0] == '0');
    OMGSR_DISPATCH_T(hipLaunchKernelGGL((attn_kernel<T, D, DMA, SPLIT>), dim3((unsigned)blocks), dim3(256), LDS, st, a, ntiles, g_defer_max, qtiles, xcd_order));
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int omgsr_set_attention_defer_max(float log2_threshold) {
    if (!(log2_threshold >= 0.0f && log2_threshold <= 12.0f)) return OMGSR_E_BADARG;
    g_defer_max = log2_threshold;
    return 0;
}

extern "C" int omgsr_attention(const omgsr_attn_args* ap, void* stream) {
    if (!ap || !ap->q || !ap->k || !ap->vt || !ap->o) return OMGSR_E_BADARG;
    const omgsr_attn_args a = *ap;
    if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0) return OMGSR_E_BADARG;
    if ((a.q_ld & 7) || (a.k_ld & 7) || (a.vt_ld & 7) || (a.o_ld & 3) || a.vt_ld < a.Lk) return OMGSR_E_SHAPE;
    if (a.o_lo_off < 0 || (a.o_lo_off && ((a.o_lo_off & 3) || a.o_lo_off < a.H * a.D || a.o_ld < (int64_t)a.o_lo_off + a.H * a.D))) return OMGSR_E_SHAPE;
    // MX output: the whole row belongs to this call (heads at column 0, o_ld = 2 H D slots), fp16 compute type
    if (a.o_mx && (a.o_lo_off || a.o_ld != 2ll * a.H * a.D || ((a.H * a.D) & 63) || omgsr::compute_dtype() != 1)) return OMGSR_E_SHAPE;
    // two-term split q / k (ABI v17): both or neither, 16-byte aligned low halves, head_dim 64 (the UNet's; FLUX's RMS-normalised q / k do not need it)
    const bool split = a.q_lo_off != 0 || a.k_lo_off != 0;
    if (split && (a.q_lo_off <= 0 || a.k_lo_off <= 0 || (a.q_lo_off & 7) || (a.k_lo_off & 7) || a.D != 64)) return OMGSR_E_SHAPE;
    if ((a.p_split || a.vt_lo_off) && !split) return OMGSR_E_SHAPE;
    if (a.vt_lo_off < 0 || (a.vt_lo_off & 7)) return OMGSR_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const double flops = 4.0 * (double)a.B * a.H * (double)a.Lq * a.Lk * a.D;         // (work handed: the split form's extra passes are overhead)
    const double bytes = 2.0 * (double)a.B * a.H * a.D * ((split ? 3.0 : 2.0) * a.Lq + (split ? 3.0 : 2.0) * a.Lk);
    omgsr::TimingScope ts(OMGSR_TK_ATTN, flops, bytes, st, (long long)a.B * a.H * a.Lq, a.Lk, a.D);
    // LDS-DMA staging needs whole 64-key tiles (a DMA piece cannot be masked) and 32-bit source offsets
    static const char* var = getenv("OMGSR_ATTN_VARIANT");      // A/B runs: "0" = register-staged K / V^T tiles everywhere
    const bool dma = (a.Lk & 63) == 0 && a.k_ld < (1 << 22) && a.vt_ld < (1 << 22) && !(var && var[0] == '0');
    if (split) return dma ? launch_attn<64, true, true>(a, st) : launch_attn<64, false, true>(a, st);
    if (a.D == 64) return dma ? launch_attn<64, true>(a, st) : launch_attn<64, false>(a, st);
    if (a.D == 128) return dma ? launch_attn<128, true>(a, st) : launch_attn<128, false>(a, st);
    return OMGSR_E_SHAPE;
}
