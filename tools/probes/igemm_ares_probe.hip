// EXPERIMENT, NOT PART OF THE LIBRARY (round 6; results: profiles/r06_experiments.md). Built, parity-green (10 kernel tests), measured SLOWER than the
// shipped kernels on every shape it targets, and removed from the dispatcher. To rebuild: copy to omgsr_amd/csrc/igemm_ares.hip, add it to
// omgsr_amd/build.py SOURCES, and in csrc/igemm.hip call igemm_ares_ok() / igemm_ares_launch() in front of the ping-pong GEMM in the LDS-DMA branch.
//
// A-in-registers GEMM for the SHORT-K token matrices of the UNet's 64 x 64 level (147 456 rows, K = 320): round 6, VERDICT r5 item 2.
//
// Why: those GEMMs are bound by bytes through the LDS-DMA path (~5 TB/s chip-wide), not by MFMA time - bytes = 2 M N K (1/BM + 1/BN) for an
// LDS-tiled kernel, i.e. the activation matrix A makes one pass per 128- or 256-column tile (3 passes at N = 320, 10-20 at the GEGLU projection's
// N = 2560). A whole row block of A is small when K is short: 32 rows x 320 channels = 20 KB = 80 registers per lane in the MFMA fragment
// layout. So every wave keeps ITS 32 rows of A in registers for the whole kernel (read once from HBM, straight into fragments, no LDS), the
// workgroup walks ALL column blocks of the problem, and only the weights stream through LDS: bytes through the DMA path = N K (M / BM) instead of
// M K (N / BN) + N K (M / BM) - at N = 320 the 283 MB of A passes disappear and 141 MB of (L2-resident) weights remain.
//
//  * workgroup = NW waves (8 or 6: BM = 256 or 192 rows; 192 makes 147 456 rows exactly three rounds of 256 CUs), wave w owns rows 32 w .. 32 w + 31
//  * column blocks of 64 (one call of the shared epilogue per block: with 128 the two calls and the 80 A registers spilled 110+ VGPRs); per block
//    the contraction runs in stages of 64 channels: a stage is a 64-row x 128-byte weight tile (8 KB, the ping-pong kernel's half-tile image:
//    16-byte slot of k-chunk c of row r at c ^ ((r >> 1) & 7)), 8 LDS-DMA pieces over the waves, 4-deep ring (prefetch distance 3) that runs on
//    seamlessly from one column block into the next; 8 MFMAs (2 column fragments x 4 k-steps) per wave and stage
//  * the weight's w_lo segment (omgsr_igemm_args.in_ld < Cin: the contraction wraps over the operand row) re-uses the SAME A registers
//  * the shared epilogue runs per 64-column half of a block (its staging region does not overlap the ring, so the next block's weights are
//    already in flight); the vmcnt queue holds epilogue loads / stores then, so the first wait after an epilogue drains it (vmcnt(0))
//  * plain 16-bit operands only (fast tiers; the accurate tier's weight-split-only layers). The MX rows (1280 bytes per 320 channels = 160
//    registers of A) do not leave room for the accumulators next to the inline-asm scaled MFMAs: not built.
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "igemm_epilogue.hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BN = 64, KST = 64;                    // columns per block, channels (16-bit slots) per stage
constexpr int STAGE_BYTES = BN * KST * 2;           // 8 KB
constexpr int NSTAGE = 4, DIST = 3, NPIECE = STAGE_BYTES / 1024;
constexpr int KREG = 320, NKF = KREG / 16;          // channels of A a wave keeps, k16 fragments
constexpr int EPI_LDW = 64 + 4;

__device__ __attribute__((aligned(16))) unsigned int g_zero_page_ar[4] = {0u, 0u, 0u, 0u};

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

constexpr int lds_bytes(int nw) { return nw * 32 * EPI_LDW * 4 + NSTAGE * STAGE_BYTES + 1024; }

// NW waves (BM = 32 NW rows); NST = Cin / 64 stages per column block (5: K = 320; 10: K = 640 = [w_hi | w_lo] over the same 320 operand channels)
template <typename T, int NW, int NST>
__global__ __launch_bounds__(NW * 64, 2) void igemm_ares_kernel(const omgsr_igemm_args p, const IgemmGeo g) {
    constexpr int BM = 32 * NW;
    constexpr int PPW = (NPIECE + NW - 1) / NW;             // DMA pieces per wave and stage (the surplus ones copy the zero page to a dummy KiB)
    constexpr int EPI_BYTES = NW * 32 * EPI_LDW * 4;
    constexpr int RING_OFF = EPI_BYTES, DUMMY_OFF = RING_OFF + NSTAGE * STAGE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int bz = blockIdx.z;
    const T* __restrict__ in = (const T*)p.in + (int64_t)bz * p.in_bstride;
    const unsigned char* wt = reinterpret_cast<const unsigned char*>((const T*)p.weight + (int64_t)bz * p.w_bstride);
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;
    const int logical_cols = (p.act == OMGSR_ACT_GEGLU) ? 2 * p.Cout : p.Cout;
    const int nblk = (logical_cols + BN - 1) / BN;
    const int total = nblk * NST;                            // stages of the whole walk
    const int64_t wrow = (int64_t)p.K_pad * 2;               // bytes per weight row

    // ---- weight stream: piece q = wave * PPW + i covers rows 8 q .. 8 q + 7 of a stage ------------------------------------------------------
    int64_t boff[PPW];
    bool breal[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        breal[i] = q < NPIECE;
        const int rl = 8 * q + (lane >> 3);
        const int chunk = (lane & 7) ^ ((4 * q + (lane >> 4)) & 7);        // (rl >> 1) & 7 == (4 q + (lane >> 4)) & 7
        boff[i] = (int64_t)rl * wrow + chunk * 16;
    }
    auto issue = [&](const int gs) {                         // global stage gs = blk * NST + s -> ring slot gs % NSTAGE
        const int blk = gs / NST, s = gs - blk * NST;
        const unsigned char* src = wt + (int64_t)blk * BN * wrow + s * (KST * 2);
        const unsigned dst = lds_base + RING_OFF + (gs % NSTAGE) * STAGE_BYTES + (wave * PPW) * 1024;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            if (breal[i]) glds16(src + boff[i], __builtin_amdgcn_readfirstlane(dst + i * 1024));
            else glds16(reinterpret_cast<const unsigned char*>(g_zero_page_ar), __builtin_amdgcn_readfirstlane(lds_base + DUMMY_OFF));
        }
    };
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < total) issue(d);

    // ---- this wave's 32 rows of A, straight into MFMA fragments (lane = row l31, k-octet 8 half of each k16 step) --------------------------
    x8_t<T> af[NKF];
    {
        const int ild = p.in_ld > 0 ? p.in_ld : p.Cin;
        int row = m0 + 32 * wave + l31;
        if (row > g.M - 1) row = g.M - 1;                    // rows past M: any valid row, the epilogue drops them
        const T* ar = in + (int64_t)row * ild + 8 * half;
#pragma unroll
        for (int ks = 0; ks < NKF; ++ks) af[ks] = *reinterpret_cast<const x8_t<T>*>(ar + 16 * ks);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // A fragments and the first stages (once per workgroup)
#pragma unroll
    for (int ks = 0; ks < NKF; ++ks) asm volatile("" : "+v"(af[ks]));

    // fragment read offsets inside a stage: column fragment cf (32 weight rows = 4 KB apart), k16 step ks
    unsigned fb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[ks] = (unsigned)(l31 * 128 + (((2 * ks + half) ^ ((l31 >> 1) & 7)) << 4));
    float* epi = reinterpret_cast<float*>(lds) + wave * 32 * EPI_LDW;

    bool drained = true;                                     // the vmcnt queue holds nothing but this wave's in-flight stages
    for (int blk = 0; blk < nblk; ++blk) {
        f32x16_t acc[1][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.0f;
        // one stage with a COMPILE-TIME index (the A fragment it meets is a register, never an indexed array: a run-time stage index sends all
        // 80 of them to scratch)
        auto stage = [&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            const int gs = blk * NST + s;
            // stage gs has landed: in steady state the DIST - 1 younger stages' pieces may still fly; right after an epilogue the queue also holds
            // its loads / stores (CDNA4 counts stores in vmcnt) - drain it once; at the end of the walk fewer stages are in flight - drain too
            if (!drained || gs + DIST > total) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            else if constexpr (PPW == 1) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            drained = true;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // slot (gs + DIST) % NSTAGE was last read in stage gs - 1: every wave passed this barrier with those reads retired
            if (gs + DIST < total) issue(gs + DIST);
            const unsigned char* Bs = lds + RING_OFF + (gs % NSTAGE) * STAGE_BYTES;
            x8_t<T> bf[4][2];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int cf = 0; cf < 2; ++cf) bf[ks][cf] = *reinterpret_cast<const x8_t<T>*>(Bs + cf * 4096 + fb[ks]);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int cf = 0; cf < 2; ++cf)
                    acc[0][cf] = mfma32(bf[ks][cf], af[(4 * s + ks) % NKF], acc[0][cf]);      // transposed tile
            __builtin_amdgcn_s_setprio(0);
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{});
        if constexpr (NST == 10) {
            stage(std::integral_constant<int, 5>{}); stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{});
            stage(std::integral_constant<int, 8>{}); stage(std::integral_constant<int, 9>{});
        }
        // the block through the shared epilogue (bias / activation / gate / residual / output form / statistics). Its arguments are laundered per
        // block: otherwise hipcc hoists ~76 registers of block-invariant address arithmetic out of the column loop and parks them in scratch
        omgsr_igemm_args q = p;
        asm volatile("" : "+s"(q.out), "+s"(q.residual), "+s"(q.bias), "+s"(q.gate), "+s"(q.out_ld), "+s"(q.out_lo_off), "+s"(q.out_bstride));
        int mrow = m0 + 32 * wave, Mq = g.M;
        asm volatile("" : "+s"(mrow), "+s"(Mq));
        igemm_epilogue_linear<T, 64, 1, 2>(q, Mq, acc, epi, lane, mrow, blk * BN, bz, q.gn_partial ? g.HoWo : 0);
        drained = false;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <typename T, int NW, int NST>
int launch_one(const omgsr_igemm_args& a, const IgemmGeo& g, hipStream_t st) {
    constexpr int LDS = lds_bytes(NW);
    static bool attr_set = false;
    if (!attr_set) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ares_kernel<T, NW, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const dim3 grid((unsigned)((g.M + 32 * NW - 1) / (32 * NW)), 1, (unsigned)a.batch);
    hipLaunchKernelGGL((igemm_ares_kernel<T, NW, NST>), grid, dim3(NW * 64), LDS, st, a, g);
    return (int)hipGetLastError();
}

}  // namespace

namespace omgsr {
// GEMM-shaped, plain 16-bit operand rows of exactly KREG channels (the contraction may wrap over them once: the weight's w_lo segment),
// 128-column weight padding, enough rows to fill the chip with 192- / 256-row workgroups, 32-bit-safe operand offsets
bool igemm_ares_ok(const omgsr_igemm_args& a, const IgemmGeo& g) {
    static const char* off = getenv("OMGSR_ARES");                 // A/B runs: "0" = never
    if (off && off[0] == '0') return false;
    const int ild = a.in_ld > 0 ? a.in_ld : a.Cin;
    return a.R == 1 && a.S == 1 && a.stride == 1 && a.pad_top == 0 && a.pad_left == 0 && a.upsample == 0 && a.Ho == a.H && a.Wo == a.W &&
           a.mx_chunks16 == 0 && !a.gn_scale_shift && ild == KREG && (a.Cin == KREG || a.Cin == 2 * KREG) && a.K_pad == a.Cin && (a.Cout_pad % BN) == 0 &&
           g.splits == 1 && (int64_t)g.M * a.batch >= 256ll * 192 && (int64_t)g.M * ild * 2 < (1ll << 32);
}

int igemm_ares_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    // rows per workgroup: 128 (4 waves, TWO workgroups per CU: one's epilogue - residual loads, LDS staging, stores: longer than a 64-column block's
    // MFMA loop - runs under the other's stages) by default; 192 / 256 (one workgroup per CU) for A/B runs: OMGSR_ARES_BM
    static const char* fb = getenv("OMGSR_ARES_BM");
    const int bm = fb ? atoi(fb) : 128;
    const bool wrap = a.Cin == 2 * KREG;
    if (bm == 192) {
        if (wrap) OMGSR_DISPATCH_T(return (launch_one<T, 6, 10>(a, g, st)));
        OMGSR_DISPATCH_T(return (launch_one<T, 6, 5>(a, g, st)));
    }
    if (bm == 256) {
        if (wrap) OMGSR_DISPATCH_T(return (launch_one<T, 8, 10>(a, g, st)));
        OMGSR_DISPATCH_T(return (launch_one<T, 8, 5>(a, g, st)));
    }
    if (wrap) OMGSR_DISPATCH_T(return (launch_one<T, 4, 10>(a, g, st)));
    OMGSR_DISPATCH_T(return (launch_one<T, 4, 5>(a, g, st)));
    return OMGSR_E_SHAPE;
}
}  // namespace omgsr
