// Mixed-precision GEMM (gfx950): the LDS-DMA implicit GEMM of igemm_dma.hip specialised to GEMM-shaped problems (1x1, stride 1: every
// Linear of the UNet's transformer blocks) whose operand arrives in the OMGSR_EL_MX form and whose weight is packed the same way:
//
//   operand row m (C logical channels, 4C bytes):   [a_hi fp16 (2C B) | a_lo' fp8 (C B) | a_hi' fp8 (C B)]
//   weight row n  (4C bytes):                       [w_hi fp16 (2C B) | w_hi' fp8 (C B) | w_lo' fp8 (C B)]
//
// A K-step is 64 bytes of every row: the first C / 32 steps are 32 fp16 channels (16 x v_mfma_f32_32x32x16_f16 per wave), the remaining
// C / 32 steps are 64 fp8 channels (8 x v_mfma_scale_f32_32x32x64_f8f6f4, 16 passes each: the same matrix-pipe time per step, twice the
// channels) whose E8M0 scale operands put the 2^-11 / 2^-s1 / 2^-s2 of the correction terms back - all into ONE fp32 accumulator:
//   a_hi w_hi  +  a_lo w_hi  +  a_hi w_lo       at 2 K segments of MFMA time and operand traffic instead of the three fp16 segments
// ([w_hi | w_hi | w_lo] x [a_hi | a_lo], third segment wrapping) the same layers ran through igemm_dma_kernel / igemm_p8_kernel.
//
// Why its own kernel: the same fp8 K-steps inside igemm_dma_kernel, behind a wave-uniform branch per K-step,
// spilled 541 VGPRs - that kernel carries the conv gather state (image / row / column of every DMA piece, tap cursor, wrap point) and
// sits at 256 registers without them. Here there is no gather (a piece's source is row * 4C + 64 k: two pointers per piece), and the
// contraction is TWO loops like the halo-tile kernel's (all fp16 steps, then all fp8 steps, the accumulators tied in place by the
// inline-asm MFMAs of the second loop). The 3-deep ring's stage of a step is step % 3 and the fp8 loop starts at stage (C / 32) % 3
// (C = 320: 10 fp16 steps -> stage 1), so the stage is a run-time LDS byte offset here and each loop is ONE straight-line body.
//
// Tile 256 (or 192) x 128, four waves of 128 x 64, two workgroups per CU, LDS image / swizzle / counted waits as in igemm_dma.hip.
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "igemm_epilogue.hip.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int BK = 32;                 // 16-bit slots per K-step = 64 bytes of a row
constexpr int BN = 128, WTN = 64, FN = 2;
constexpr int NSTAGE = 3;
constexpr int lds_bytes(int bm) { return NSTAGE * (bm * 64 + BN * 64); }
typedef _Float16 T;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

OMGSR_DEVINL void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}

template <int BM>
__global__ __launch_bounds__(256, 2) void igemm_gmx_kernel(const omgsr_igemm_args p, const IgemmGeo g) {
    constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int WTM = BM / 2, FM = WTM / 32;
    constexpr int APW = (BM / 16) / 4, BPW = (BN / 16) / 4;          // 1-KiB DMA pieces per wave per K-step
    constexpr int PIECES = APW + BPW;
    static_assert(APW * 64 == BM && (WTM % 32) == 0, "tile / wave split");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // same L2-aware tile order as igemm_dma_kernel: per XCD a contiguous id range walking 8 m-tiles before advancing n
    const int tile = xcd_remap(blockIdx.x, g.ntm * g.ntn);
    const int per_group = 8 * g.ntn;
    const int grp = tile / per_group, in_grp = tile - grp * per_group;
    const int first_m = grp * 8;
    const int gsz = (g.ntm - first_m) < 8 ? (g.ntm - first_m) : 8;
    const int tm = first_m + in_grp % gsz, tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;
    const int64_t row_bytes = (int64_t)p.Cin * 2;                    // Cin counts 16-bit slots: 2C of them = 4C bytes

    // ---- DMA coordinates: a wave instruction fills 16 rows x 64 B; rows past M read row M - 1 (the epilogue drops them) ------------
    const int lrow = lane >> 2;
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);                   // source chunk for LDS position (lane & 3)
    const unsigned char* a_ptr[APW];
    const unsigned char* b_ptr[BPW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        int m = m0 + 16 * (wave * APW + i) + lrow;
        m = m < g.M ? m : g.M - 1;
        a_ptr[i] = reinterpret_cast<const unsigned char*>(p.in) + (int64_t)m * row_bytes + kc * 16;
    }
#pragma unroll
    for (int i = 0; i < BPW; ++i)
        b_ptr[i] = reinterpret_cast<const unsigned char*>(p.weight) + (int64_t)(n0 + 16 * (wave * BPW + i) + lrow) * p.K_pad * 2 + kc * 16;

    auto issue = [&](const int stage_off) {          // stage_off = ring stage * STAGE_BYTES (wave-uniform)
        const unsigned sa = lds_base + stage_off + (16 * wave * APW) * 64;
        const unsigned sb = lds_base + stage_off + A_BYTES + (16 * wave * BPW) * 64;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            glds16(a_ptr[i], __builtin_amdgcn_readfirstlane(sa + i * 1024));
            a_ptr[i] += 64;
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            glds16(b_ptr[i], __builtin_amdgcn_readfirstlane(sb + i * 1024));
            b_ptr[i] += 64;
        }
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.K_pad / BK;             // all K-steps
    const int n16 = p.mx_chunks16;           // the fp16 ones (the rest are fp8)
    const int seg2_at = n16 + (nk - n16) / 2;        // first step of the a_hi' x w_lo' segment
    issue(0);
    if (nk > 1) issue(STAGE_BYTES);

    // fragment read offsets (swizzled): row = base + (lane & 31); 16-byte slot = (2 ks + (lane >> 5)) ^ ((row >> 2) & 3)
    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    const int foff0 = frow * 64 + (((lane >> 5)) ^ fsw) * 16;
    const int foff1 = frow * 64 + ((2 + (lane >> 5)) ^ fsw) * 16;
    const unsigned char* fa0 = lds + (wm * WTM) * 64 + foff0;
    const unsigned char* fa1 = lds + (wm * WTM) * 64 + foff1;
    const unsigned char* fb0 = lds + A_BYTES + (wn * WTN) * 64 + foff0;
    const unsigned char* fb1 = lds + A_BYTES + (wn * WTN) * 64 + foff1;

    // one K-step; the ring stage is a RUN-TIME byte offset (four v_add per step on the fragment bases, SALU on the DMA destinations)
    // so that each loop has ONE body: with compile-time stages the fp8 loop needs three bodies behind a wave-uniform branch (its first
    // stage is (C / 32) % 3), and accumulators that flow through inline-asm MFMAs in divergent paths cost 160-290 spilled VGPRs.
    // F8: the step holds 64 fp8 channels per row - the same 64-byte rows and fragment addresses; a lane's two 16-byte reads are the 32
    // consecutive k of its k-block (A and B are read at the same byte positions, so the products pair up whatever the byte order inside
    // a row means), one instruction of 16 passes where the fp16 path issues two of 8.
    auto kstep = [&](auto f8_c, const int kt, const int soff, const int soff_next2) {
        constexpr bool F8 = decltype(f8_c)::value;
        // only what the PREVIOUS step issued may still be in flight; lgkmcnt(0): this wave's fragment reads of the previous step have
        // returned before it passes the barrier that lets the others overwrite that stage (see igemm_halo_body.hip.h)
        if (kt + 1 < nk) {
            static_assert(PIECES >= 3 && PIECES <= 6, "counted wait");
            if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else if constexpr (PIECES == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
            else if constexpr (PIECES == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned char* a0 = fa0 + soff; const unsigned char* a1 = fa1 + soff;
        const unsigned char* b0 = fb0 + soff; const unsigned char* b1 = fb1 + soff;
        if constexpr (F8) {
            // scale operands: VGPRs written by VALU moves in front of the fragment reads (the asm MFMAs that read them are invisible to
            // the compiler's hazard padding; the LDS reads and their wait put >> the 2 required wait states in between)
            int sw = kt < seg2_at ? p.mx_scale_w1 : p.mx_scale_w2, sa = kt < seg2_at ? p.mx_scale_a1 : p.mx_scale_a2;
            asm volatile("" : "+v"(sw), "+v"(sa));
            i32x8_t a8[FM], b8[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i)
                a8[i] = __builtin_shufflevector(*reinterpret_cast<const i32x4_t*>(a0 + i * 32 * 64), *reinterpret_cast<const i32x4_t*>(a1 + i * 32 * 64),
                                                0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                b8[j] = __builtin_shufflevector(*reinterpret_cast<const i32x4_t*>(b0 + j * 32 * 64), *reinterpret_cast<const i32x4_t*>(b1 + j * 32 * 64),
                                                0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    // inline asm with the accumulator tied in place: through the builtin the register allocator moves whole accumulators
                    // between the fp16 and the fp8 loop. Dependent MFMAs on one accumulator are 7 instructions of 16 passes apart.
                    asm volatile("s_nop 3\n\tv_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                                 : "+v"(acc[i][j]) : "v"(b8[j]), "v"(a8[i]), "v"(sw), "v"(sa));   // transposed tile
                if (i == (FM > 1 ? FM / 2 - 1 : 0)) {               // this step's LDS-DMA pieces behind the first half of its MFMAs (one row block: behind all of them)
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 2 < nk) issue(soff_next2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            x8_t<T> af[2][FM], bf[2][FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                af[0][i] = *reinterpret_cast<const x8_t<T>*>(a0 + i * 32 * 64);
                af[1][i] = *reinterpret_cast<const x8_t<T>*>(a1 + i * 32 * 64);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bf[0][j] = *reinterpret_cast<const x8_t<T>*>(b0 + j * 32 * 64);
                bf[1][j] = *reinterpret_cast<const x8_t<T>*>(b1 + j * 32 * 64);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mfma32(bf[ks][j], af[ks][i], acc[i][j]);   // transposed tile
                if (ks == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 2 < nk) issue(soff_next2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };
    // ring stage of step kt = kt % 3 in both loops: soff walks 0, STAGE_BYTES, 2 STAGE_BYTES; the stage refilled in step kt (for step
    // kt + 2) is the one before it in that cycle
    int soff = 0, kt = 0;
    auto advance = [&]() { soff = soff == 2 * STAGE_BYTES ? 0 : soff + STAGE_BYTES; };
    auto prev_of = [&](const int so) { return so == 0 ? 2 * STAGE_BYTES : so - STAGE_BYTES; };
    for (; kt < n16; ++kt) {
        kstep(std::false_type{}, kt, soff, prev_of(soff));
        advance();
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                 // (the asm MFMAs below are invisible to the compiler's hazard padding)
    for (; kt < nk; ++kt) {
        kstep(std::true_type{}, kt, soff, prev_of(soff));
        advance();
    }
    // the last asm MFMAs (16 passes each) must have written the accumulators before the epilogue's VALU reads them
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    static_assert(4 * 32 * (WTN + 4) * 4 <= lds_bytes(BM), "epilogue staging must fit the allocation");
    float* epi = reinterpret_cast<float*>(lds) + wave * 32 * (WTN + 4);
    igemm_epilogue_linear<T, WTN, FM, FN>(p, g.M, acc, epi, lane, m0 + wm * WTM, n0 + wn * WTN, 0, p.gn_partial ? g.HoWo : 0);
}

template <int BM>
int launch_gmx(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    g.ntm = (g.M + BM - 1) / BM;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    g.ntn = (logical_cols + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_gmx_kernel<BM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(BM));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((igemm_gmx_kernel<BM>), dim3(g.ntm * g.ntn), dim3(256), lds_bytes(BM), st, a, g);
    return (int)hipGetLastError();
}

}  // namespace

namespace omgsr {
int compute_dtype();
// GEMM-shaped problem over an OMGSR_EL_MX operand: 1x1, stride 1, no padding / upsampling, one batch entry, C % 64 == 0 logical channels
// (Cin = 2C 16-bit slots), fp16 compute type, no wrap (the row carries its own fp8 copies)
bool igemm_gmx_ok(const omgsr_igemm_args& a) {
    return a.mx_chunks16 > 0 && a.R == 1 && a.S == 1 && a.stride == 1 && a.pad_top == 0 && a.pad_left == 0 && a.upsample == 0 && a.Ho == a.H && a.Wo == a.W &&
           a.batch == 1 && a.in_ld == 0 && (a.Cin % 128) == 0 && a.K_pad == a.Cin && a.mx_chunks16 == a.Cin / 64 && compute_dtype() == 1;
}
int igemm_gmx_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    // tile-count quantisation on the 512 slots (2 workgroups per CU), as igemm_dma_launch: 192 rows when the 256-row grid leaves the last
    // round under ~3/4 full and the 192-row grid fills it better
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    const int64_t nt = (logical_cols + 127) / 128;
    const int64_t t256 = (int64_t)((g.M + 255) / 256) * nt, t192 = (int64_t)((g.M + 191) / 192) * nt;
    auto eff = [](int64_t tiles) { const int64_t rounds = (tiles + 511) / 512; return (double)tiles / (double)(rounds * 512); };
    static const char* bm = getenv("OMGSR_GMX_BM");              // A/B runs: "256" | "192" | "128" | "64"
    const bool force192 = bm && !strcmp(bm, "192"), force256 = bm && !strcmp(bm, "256");
    // Small-M regime (round 5: one image per call, M = 64 ... 4096 token rows): a handful of 256-row tiles leaves most CUs idle AND makes every
    // K-step of the few busy ones 16 MFMAs long on one wave per SIMD. 64- / 128-row tiles (one / two 32-row fragments per wave) shorten the
    // step and quadruple the workgroups; everything is L2-resident at this size, so the extra weight reads cost nothing.
    if (!force256 && !force192) {
        if ((bm && !strcmp(bm, "64")) || (!bm && t256 < 64)) return launch_gmx<64>(a, g, st);
        if ((bm && !strcmp(bm, "128")) || (!bm && t256 < 160)) return launch_gmx<128>(a, g, st);
    }
    if (!force256 && (force192 || (eff(t256) < 0.78 && eff(t192) > eff(t256) + 0.1))) return launch_gmx<192>(a, g, st);
    return launch_gmx<256>(a, g, st);
}
}  // namespace omgsr
