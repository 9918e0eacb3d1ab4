// GEMM-shaped implicit GEMM for the big token matrices (Flux linears, the UNet's wide projections): 256 x 256 tile, BK 64,
// 8 waves in two groups that run half a phase apart ("ping-pong"): while one group's four waves (one per SIMD) issue their
// fragment reads and LDS-DMA prefetches, the other group's four run MFMAs, so the matrix pipe of every SIMD always has exactly
// one wave feeding it and every LDS / L2 latency of the loading wave sits behind its partner's MFMA cluster.
// (igemm_dma_kernel's 8 waves move in lockstep: all read, then all multiply - its operand stream and its compute loop each take
// as long alone as this kernel takes for both, profiles/r02_dma_ablation.md.)
//
// Structure (CDNA4 playbook, "256^2 8-phase" schedule, restated for the transposed 32x32x16 accumulators of this library):
//  * LDS: two K-tile buffers (E, O) x four 16 KB half-tiles (A rows 0-127 / 128-255, B rows 0-127 / 128-255; a row = 64 k = 128 B)
//    + 1 KB that swallows the DMA pieces issued past the end of K (every phase issues exactly two pieces per wave, so the counted
//    vmcnt waits never change). Half-tiles land by LDS-DMA (8 rows x 128 B per wave instruction); the 16-byte slot of k-chunk c
//    of row r is c ^ ((r >> 1) & 7), applied to the SOURCE address and to the fragment reads: conflict-free ds_read_b128.
//  * a K-tile is consumed in four phases, one 64 x 32 quadrant of the wave's 128 x 64 output each (8 MFMAs 32x32x16, K = 64):
//    (rows 0-63, cols 0-31) -> (0-63, 32-63) -> (64-127, 32-63) -> (64-127, 0-31); the fragments read in front of them are
//    B0 + A0 (12 x ds_read_b128), B1 (4), A1 (8), none (B0 is still in registers): all reads of a buffer fall in its first three
//    phases, its B half-tiles are dead after the second and its A half-tiles after the third.
//  * phase = [fragment reads, one half-tile of prefetch (2 pieces), s_waitcnt lgkmcnt(0) (+ vmcnt(4) in the 4th / 8th phase)]
//    s_barrier [s_setprio 1, 8 MFMAs, s_setprio 0] s_barrier. Group 1 executes one extra s_barrier in front of the loop (group 0
//    one behind it), which is the half-phase stagger.
//  * prefetch order over the 8 phases of an iteration (tiles t in E, t + 1 in O): O.A0, O.A1 (t + 1) | E.B0, E.B1 (t + 2) |
//    E.A0, E.A1 (t + 2) | O.B0, O.B1 (t + 3): each slot is refilled >= 1 phase after its last read retired (lgkmcnt(0) sits in
//    front of the barrier) and is waited for (vmcnt(4): only the two youngest half-tiles may still fly) one phase before its first
//    read - with two staggered groups a wait in phase w is visible to every reader from phase w + 1 on.
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "igemm_epilogue.hip.h"
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;           // 16 KB
constexpr int BUF_BYTES = 4 * HALF_BYTES;          // A0 A1 B0 B1
constexpr int DUMMY_OFF = 2 * BUF_BYTES;
constexpr int LDS_BYTES = DUMMY_OFF + 1024;
constexpr int WTN = 64, FM = 4, FN = 2;

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

// MX (round 4, T = fp16 only): the operand / weight rows are OMGSR_EL_MX rows (see igemm_gmx.hip): the first mx_chunks16 / 2 K-tiles hold 64
// fp16 slots, the others 128 fp8 channels - two v_mfma_scale_f32_32x32x64_f8f6f4 per fragment pair instead of four fp16 MFMAs, the same 256
// matrix-pipe cycles per phase. K-tiles come in (E, O) pairs, so the contraction is three loops: fp16 pairs, ONE mixed pair when the number
// of fp16 K-tiles is odd (C = 320: 5 + 5 K-tiles), fp8 pairs. Fragments live in i32x8 tuples whose halves the fp16 phases use as they are.
template <typename T, bool MX = false>
__global__ __launch_bounds__(512, 2) void igemm_p8_kernel(const omgsr_igemm_args p, const IgemmGeo g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wr = wave >> 2, wc = wave & 3;           // wr: the wave's 128 rows = A half-tile wr; also its ping-pong group
    const int half = lane >> 5, l31 = lane & 31;

    // same L2-aware tile order as igemm_dma_kernel: per XCD a contiguous id range walking 8 m-tiles before advancing n
    const int tile = xcd_remap(blockIdx.x, g.ntm * g.ntn);
    const int per_group = 8 * g.ntn;
    const int grp = tile / per_group, in_grp = tile - grp * per_group;
    const int first_m = grp * 8;
    const int gsz = (g.ntm - first_m) < 8 ? (g.ntm - first_m) : 8;
    const int tm = first_m + in_grp % gsz, tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.z;
    const unsigned char* abase = reinterpret_cast<const unsigned char*>((const T*)p.in + (int64_t)bz * p.in_bstride);
    const unsigned char* bbase = reinterpret_cast<const unsigned char*>((const T*)p.weight + (int64_t)bz * p.w_bstride);
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;
    const int nkt = p.Cin / BK;                        // even, >= 2 (igemm_p8_ok)
    const int ild = p.in_ld > 0 ? p.in_ld : p.Cin;     // physical channels per operand row
    const int wrap_at = ild / BK;                      // K-tiles at / past this one re-read the operand row from its start (w_lo segment)

    // ---- prefetch coordinates: piece j (0, 1) of wave w covers rows 16 w + 8 j .. + 7 of a half-tile ----------------------
    unsigned aoff[2][2], boff[2][2];                   // [half-tile][piece]: byte offsets off abase / bbase (k-tile 0)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int rl = 16 * wave + 8 * j + (lane >> 3);
            const int chunk = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7);          // (rl >> 1) & 7 == (4 j + (lane >> 4)) & 7
            int m = m0 + 128 * h + rl;
            if (m > g.M - 1) m = g.M - 1;                                         // rows past M: any valid row, the epilogue drops them
            aoff[h][j] = (unsigned)(((int64_t)m * ild + chunk * 8) * 2);
            boff[h][j] = (unsigned)(((int64_t)(n0 + 128 * h + rl) * p.K_pad + chunk * 8) * 2);
        }
    // slot: 0, 1 = A half-tiles, 2, 3 = B half-tiles; kt past the end -> the dummy KiB (count stays 2 pieces per phase)
    auto stage = [&](const int buf, const int slot, const int kt) {
        const bool real = kt < nkt;
        const int k = real ? kt : nkt - 1;
        const unsigned char* sb = slot < 2 ? abase + (int64_t)(k >= wrap_at ? k - wrap_at : k) * (BK * 2) : bbase + (int64_t)k * (BK * 2);
        const unsigned dst = lds_base + buf * BUF_BYTES + slot * HALF_BYTES + (2 * wave) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned vo = slot == 0 ? aoff[0][j] : slot == 1 ? aoff[1][j] : slot == 2 ? boff[0][j] : boff[1][j];
            glds16_sv(vo, sb, __builtin_amdgcn_readfirstlane(real ? dst + j * 1024 : lds_base + DUMMY_OFF));
        }
    };

    // ---- fragment read offsets (per lane, per buffer, per 16-wide k-step): row l31 of a 32-row block, chunk 2 ks + half -----
    unsigned fa[2][4], fb[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const unsigned f = (unsigned)(l31 * 128 + (((2 * ks + half) ^ ((l31 >> 1) & 7)) << 4));
            fa[x][ks] = f + x * BUF_BYTES + wr * HALF_BYTES;
            fb[x][ks] = f + x * BUF_BYTES + (2 + (wc >> 1)) * HALF_BYTES + (wc & 1) * (64 * 128);
        }

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- prologue: tile 0 into E, the B half-tiles of tile 1 into O (its A half-tiles follow in phases 0, 1) --------------
    stage(0, 0, 0); stage(0, 1, 0); stage(0, 2, 0); stage(0, 3, 0);
    stage(1, 2, 1); stage(1, 3, 1);
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (wr == 1) __builtin_amdgcn_s_barrier();         // group 1 runs half a phase behind group 0 from here on

    if constexpr (MX) {
        typedef int i32x4_t __attribute__((ext_vector_type(4)));
        i32x8_t a8[2][2], b08[2], b18[2];              // [ii][i] / [i]: k-steps 2 i (low half) and 2 i + 1 (high half) of a K-tile
        const int n16t = p.mx_chunks16 >> 1;           // fp16 K-tiles
        const int cbytes = p.Cin >> 1;                 // C: bytes of each fp8 segment of a row (a_lo' | a_hi')
        auto ld8 = [&](const unsigned lo, const unsigned hi) {
            return __builtin_shufflevector(*reinterpret_cast<const i32x4_t*>(lds + lo), *reinterpret_cast<const i32x4_t*>(lds + hi), 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto lo4 = [](const i32x8_t v) { return __builtin_bit_cast(x8_t<T>, __builtin_shufflevector(v, v, 0, 1, 2, 3)); };
        auto hi4 = [](const i32x8_t v) { return __builtin_bit_cast(x8_t<T>, __builtin_shufflevector(v, v, 4, 5, 6, 7)); };
        auto phase = [&](auto P_c, auto f8_c, const int kt) {
            constexpr int P = decltype(P_c)::value, X = P >> 2, q = P & 3;
            constexpr bool F8 = decltype(f8_c)::value;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (q == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) b08[i] = ld8(fb[X][2 * i], fb[X][2 * i + 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int i = 0; i < 2; ++i) a8[ii][i] = ld8(fa[X][2 * i] + ii * 4096, fa[X][2 * i + 1] + ii * 4096);
            } else if constexpr (q == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) b18[i] = ld8(fb[X][2 * i] + 4096, fb[X][2 * i + 1] + 4096);
            } else if constexpr (q == 2) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int i = 0; i < 2; ++i) a8[ii][i] = ld8(fa[X][2 * i] + (2 + ii) * 4096, fa[X][2 * i + 1] + (2 + ii) * 4096);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (P == 0) stage(1, 0, kt + 1);
            else if constexpr (P == 1) stage(1, 1, kt + 1);
            else if constexpr (P == 2) stage(0, 2, kt + 2);
            else if constexpr (P == 3) stage(0, 3, kt + 2);
            else if constexpr (P == 4) stage(0, 0, kt + 2);
            else if constexpr (P == 5) stage(0, 1, kt + 2);
            else if constexpr (P == 6) stage(1, 2, kt + 3);
            else stage(1, 3, kt + 3);
            // E8M0 scale operands of this K-tile's two 64-byte halves: a half belongs to the a_lo' x w_hi' segment while its byte offset in the
            // fp8 part of the row is below C, to a_hi' x w_lo' from there on (C = 320: the third fp8 K-tile straddles the two). VGPRs written
            // by VALU moves in front of the waits (the asm MFMAs that read them are invisible to the compiler's hazard padding).
            int sw[2] = {0, 0}, sa[2] = {0, 0};
            if constexpr (F8) {
                const int off = (kt + X - n16t) * 128;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bool seg2 = off + 64 * i >= cbytes;
                    sw[i] = seg2 ? p.mx_scale_w2 : p.mx_scale_w1;
                    sa[i] = seg2 ? p.mx_scale_a2 : p.mx_scale_a1;
                    asm volatile("" : "+v"(sw[i]), "+v"(sa[i]));
                }
            }
            if constexpr (q == 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            constexpr int i0 = (q >= 2) ? 2 : 0;
            constexpr int j = (q == 1 || q == 2) ? 1 : 0;
            if constexpr (F8) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii)
                        asm volatile("s_nop 3\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                                     : "+v"(acc[i0 + ii][j]) : "v"(j ? b18[i] : b08[i]), "v"(a8[ii][i]), "v"(sw[i]), "v"(sa[i]));   // transposed tile
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) acc[i0 + ii][j] = mfma32(lo4(j ? b18[i] : b08[i]), lo4(a8[ii][i]), acc[i0 + ii][j]);
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) acc[i0 + ii][j] = mfma32(hi4(j ? b18[i] : b08[i]), hi4(a8[ii][i]), acc[i0 + ii][j]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        auto pair = [&](auto fe_c, auto fo_c, const int kt) {
            phase(std::integral_constant<int, 0>{}, fe_c, kt);
            phase(std::integral_constant<int, 1>{}, fe_c, kt);
            phase(std::integral_constant<int, 2>{}, fe_c, kt);
            phase(std::integral_constant<int, 3>{}, fe_c, kt);
            phase(std::integral_constant<int, 4>{}, fo_c, kt);
            phase(std::integral_constant<int, 5>{}, fo_c, kt);
            phase(std::integral_constant<int, 6>{}, fo_c, kt);
            phase(std::integral_constant<int, 7>{}, fo_c, kt);
        };
        int kt = 0;
        for (; kt + 1 < n16t; kt += 2) pair(std::false_type{}, std::false_type{}, kt);
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");        // (the asm MFMAs below are invisible to the compiler's hazard padding)
        if (kt < n16t) { pair(std::false_type{}, std::true_type{}, kt); kt += 2; }
        for (; kt < nkt; kt += 2) pair(std::true_type{}, std::true_type{}, kt);
        // the last asm MFMAs (16 passes each) must have written the accumulators before the epilogue's VALU reads them
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    } else {
    x8_t<T> af[2][4], b0[4], b1[4];
    auto phase = [&](auto P_c, const int kt) {
        constexpr int P = decltype(P_c)::value, X = P >> 2, q = P & 3;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (q == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) b0[ks] = *reinterpret_cast<const x8_t<T>*>(lds + fb[X][ks]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af[ii][ks] = *reinterpret_cast<const x8_t<T>*>(lds + fa[X][ks] + ii * 4096);
        } else if constexpr (q == 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) b1[ks] = *reinterpret_cast<const x8_t<T>*>(lds + fb[X][ks] + 4096);
        } else if constexpr (q == 2) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) af[ii][ks] = *reinterpret_cast<const x8_t<T>*>(lds + fa[X][ks] + (2 + ii) * 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
        // one half-tile of prefetch per phase
        if constexpr (P == 0) stage(1, 0, kt + 1);
        else if constexpr (P == 1) stage(1, 1, kt + 1);
        else if constexpr (P == 2) stage(0, 2, kt + 2);
        else if constexpr (P == 3) stage(0, 3, kt + 2);
        else if constexpr (P == 4) stage(0, 0, kt + 2);
        else if constexpr (P == 5) stage(0, 1, kt + 2);
        else if constexpr (P == 6) stage(1, 2, kt + 3);
        else stage(1, 3, kt + 3);
        // own fragment reads retired (so the slot may be refilled one phase later); 4th / 8th phase: everything but the two
        // youngest half-tiles has landed = the other buffer is complete, it is read from the next phase on
        if constexpr (q == 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        constexpr int i0 = (q >= 2) ? 2 : 0;
        constexpr int j = (q == 1 || q == 2) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ii = 0; ii < 2; ++ii)
                acc[i0 + ii][j] = mfma32(j ? b1[ks] : b0[ks], af[ii][ks], acc[i0 + ii][j]);          // transposed tile
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        phase(std::integral_constant<int, 0>{}, kt);
        phase(std::integral_constant<int, 1>{}, kt);
        phase(std::integral_constant<int, 2>{}, kt);
        phase(std::integral_constant<int, 3>{}, kt);
        phase(std::integral_constant<int, 4>{}, kt);
        phase(std::integral_constant<int, 5>{}, kt);
        phase(std::integral_constant<int, 6>{}, kt);
        phase(std::integral_constant<int, 7>{}, kt);
    }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();         // re-align the groups
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");        // the dummy pieces of the last phases

    static_assert(8 * 32 * (WTN + 4) * 4 <= LDS_BYTES, "epilogue staging must fit the allocation");
    float* epi = reinterpret_cast<float*>(lds) + wave * 32 * (WTN + 4);
    igemm_epilogue_linear<T, WTN, FM, FN>(p, g.M, acc, epi, lane, m0 + wr * 128, n0 + wc * WTN, bz, p.gn_partial ? g.HoWo : 0);
}

}  // namespace

namespace omgsr {
// GEMM-shaped problems only (1x1, stride 1, no padding / upsampling: row m of the operand is in + m * Cin), whole 64-deep K-tiles
// in pairs, 256-column weight padding, 32-bit operand offsets.
bool igemm_p8_ok(const omgsr_igemm_args& a, const IgemmGeo& g) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    return a.R == 1 && a.S == 1 && a.stride == 1 && a.pad_top == 0 && a.pad_left == 0 && a.upsample == 0 && a.Ho == a.H && a.Wo == a.W &&
           a.K_pad == a.Cin && (a.Cin % (2 * BK)) == 0 && (a.in_ld == 0 || (a.in_ld % BK) == 0) && (a.Cout_pad % BN) == 0 && logical_cols >= BN && g.splits == 1 &&
           (int64_t)g.M * a.Cin * 2 < (1ll << 32) && (int64_t)a.Cout_pad * a.K_pad * 2 < (1ll << 32);
}
// ... and worth it: enough 256 x 256 tiles to fill the chip, no more padded columns than the 128-wide grid would compute
bool igemm_p8_wanted(const omgsr_igemm_args& a, const IgemmGeo& g) {
    if (!igemm_p8_ok(a, g)) return false;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    const int64_t t256 = (int64_t)((g.M + BM - 1) / BM) * ((logical_cols + BN - 1) / BN) * a.batch;
    const int cols256 = ((logical_cols + 255) / 256) * 256, cols128 = ((logical_cols + 127) / 128) * 128;
    // (round 2: K >= 1536, because with fewer K-tiles the six half-tiles of prologue and the 129 KB of LDS (one workgroup per CU) cost more
    // than the schedule wins: the UNet's GEGLU projections, K = 640 / 1280: S-1024 step 165.1 -> 165.7 ms with them on this kernel)
    static const char* mt = getenv("OMGSR_P8_MIN_TILES");          // A/B runs
    const int min_tiles = mt ? atoi(mt) : 128;      // half the CUs: 192 tiles (Flux context tokens, M = 4096) run 25 % faster here than as 384 tiles of 256 x 128
    // K threshold: round 2 set 1536 (shorter K loses to the exposed prologue / epilogue of one workgroup per CU). Round 4 measured why the
    // 256 x 128 tile is slow on the same problems - its operand stream moves 1.5x the bytes through the LDS-DMA path, which is what bounds a
    // short-K GEMM (profiles/r04_experiments.md) - and re-ran the A/B: K >= 640 is 0.3-0.8 % faster on both tiers' S-1024 steps
    // (bf16 130.1 -> 129.1 ms: 52 launches move, dma -5.6 ms, p8 +5.0 ms), K >= 1280 is in between. OMGSR_P8_MIN_K for A/B.
    static const char* mk = getenv("OMGSR_P8_MIN_K");
    const int min_k = mk ? atoi(mk) : 10 * BK;
    // padded columns allowed, in 16ths of the 128-wide grid's: 20 since round 6 (N = 640 -> 768 columns of 256-wide tiles: three passes of the operand
    // through the LDS-DMA path instead of five outweigh 20 % more MFMA work on these stream-bound problems: 44 launches of the S-1024 step move,
    // gmx -5.4 ms, dma -1.1 ms, p8 +5.8 ms; one-box A/B 190.30 -> 189.52 ms accurate, 120.04 -> 119.92 bf16). OMGSR_P8_PAD_NUM=17 = round 5.
    static const char* pp = getenv("OMGSR_P8_PAD_NUM");
    const int pad_num = pp ? atoi(pp) : 20;
    return t256 >= min_tiles && cols256 * 16 <= cols128 * pad_num && a.Cin >= min_k;
}

int igemm_p8_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    g.ntm = (g.M + BM - 1) / BM;
    g.ntn = (logical_cols + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_p8_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_p8_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(g.ntm * g.ntn, 1, a.batch);
    if (a.mx_chunks16 > 0) {            // mixed-precision rows (fp16 compute type; the dispatcher checked igemm_gmx_ok)
        static bool mx_attr = false;
        if (!mx_attr) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_p8_kernel<f16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
            if (e != hipSuccess) return (int)e;
            mx_attr = true;
        }
        hipLaunchKernelGGL((igemm_p8_kernel<f16_t, true>), grid, dim3(512), LDS_BYTES, st, a, g);
        return (int)hipGetLastError();
    }
    OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_p8_kernel<T>), grid, dim3(512), LDS_BYTES, st, a, g));
    return (int)hipGetLastError();
}
}  // namespace omgsr
