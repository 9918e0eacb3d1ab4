// Implicit-GEMM conv / linear, LDS-DMA variant for large problems (gfx950).
//
// Why a second kernel: the register-staged kernel (igemm.hip) pays one ds_write_b128 per staged
// 16 bytes, and gfx950's register->LDS path sustains only ~79 B/clk/CU (MI355X_MICROARCH.md §LDS):
// at a 128x128 tile that is ~0.8 LDS-write cycles per MFMA cycle, which caps the kernel near 30 % of
// the MFMA peak. Here the operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
// instruction, no VGPR round trip, no ds_write), the tile is 256 x 128 and three K-stages are in flight.
//
//  * LDS image is LINEAR ([row][64 B], BK = 32) because the DMA writes base + lane*16; bank
//    conflicts are removed by a swizzle applied to the SOURCE chunk and to the read
//    (position = chunk ^ ((row >> 2) & 3)): every 16-lane ds_read_b128 service group then covers
//    16 distinct 16-B slots
//  * the conv halo / rows past M read a 16-byte ZERO PAGE instead of being predicated (an LDS-DMA lane
//    that is masked off would leave stale bytes in its slot)
//  * K order is tap-major (Cin % 32 == 0): validity + pointer of each gathered row are recomputed once
//    per TAP; inside a tap a K-step only advances pointers (the first version was instruction-issue
//    bound: ~10 VALU + 8 SALU per MFMA, profiles/r01_pmc_igemm.md)
//  * 3-deep LDS ring, prefetch distance 2, ONE raw s_barrier per K-step; waits are counted
//    (s_waitcnt vmcnt(P) = "my copies for this step have landed, next step's may still fly");
//    the DMA is issued from inline asm so hipcc neither drains it at the barrier nor before ds_reads
//  * two wave shapes: WGM = 4 -> 8 waves of 64x64 (1 ds_read_b128 per MFMA, 8 MFMA per barrier),
//    WGM = 2 -> 4 waves of 128x64 (0.75 reads per MFMA, 16 MFMA per barrier, 128 accumulator VGPRs)
//  * epilogue shared with igemm.hip
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "igemm_epilogue.hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 32;
constexpr int WTN = 64, FN = 2;
constexpr int NSTAGE = 3;
// BM = 256 (default) or 192 (tile-count quantisation: M = 9216 rows x 1280 columns is 360 tiles of 256 x 128 on 512
// workgroup slots, 480 of 192 x 128). BN = 64 * WGN: 128 (24 KB / stage, 12 KB of operands per MFLOP) or 256 (32 KB, 8 KB).
constexpr int lds_bytes(int bm, int wgn) { return NSTAGE * (bm * BK * 2 + wgn * 64 * BK * 2); }

__device__ __attribute__((aligned(16))) unsigned int g_zero_page[4] = {0u, 0u, 0u, 0u};

// one LDS-DMA piece: 64 lanes x 16 B -> LDS [lds_dst, lds_dst + 1 KiB); lds_dst is wave-uniform.
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

template <int N>
OMGSR_DEVINL void wait_vmcnt() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else static_assert(N == 0, "unsupported count");
}

// ABL (ablation, A/B runs only): 0 = the kernel; 1 = no DMA; 2 = DMA + ds_reads, no MFMA; 3 = DMA only; 4 = no epilogue;
// 5 = LDS-DMA issued in front of the step's MFMAs; 6 = no fragment reads (profiles/r02_dma_ablation.md)
template <typename T, int WGM, int WGN, int ABL = 0, int BM = 256>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void igemm_dma_kernel(const omgsr_igemm_args p, const IgemmGeo g) {
    constexpr int A_BYTES = BM * BK * 2;       // 16 KB at BM 256
    constexpr int NW = WGM * WGN;              // waves
    constexpr int BN = WGN * WTN;
    constexpr int B_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int WTM = BM / WGM, FM = WTM / 32;
    constexpr int APW = (BM / 16) / NW, BPW = (BN / 16) / NW; // 1-KiB DMA pieces per wave per K-step (A: BM/16, B: BN/16 in total)
    static_assert(APW * NW * 16 == BM && (BM / WGM) % 32 == 0, "tile / wave split");
    constexpr int PIECES = APW + BPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // LDS_BYTES, dynamic

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    // L2-aware order: the XCD remap hands each XCD a contiguous id range; inside it ids walk 8 m-tiles before
    // advancing n, so the ~64 workgroups resident on an XCD form an 8 x 8 super-tile sharing 8 A and 8 B panels
    // (n-fastest order shared ONE A panel and streamed 64 distinct B panels: 14x the ideal HBM traffic on the
    // Flux linears, profiles/r01_pmc_igemm.md §3)
    const int tile = xcd_remap(blockIdx.x, g.ntm * g.ntn);
    const int per_group = 8 * g.ntn;
    const int grp = tile / per_group, in_grp = tile - grp * per_group;
    const int first_m = grp * 8;
    const int gsz = (g.ntm - first_m) < 8 ? (g.ntm - first_m) : 8;
    const int tm = first_m + in_grp % gsz, tn = in_grp / gsz;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.z;

    const T* __restrict__ in = (const T*)p.in + (int64_t)bz * p.in_bstride;
    const T* __restrict__ wt = (const T*)p.weight + (int64_t)bz * p.w_bstride;
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;            // LDS byte offset of the ring

    // ---- DMA coordinates: a wave instruction fills 16 rows x 64 B ---------------------------
    // A piece j of wave w covers rows [16*(w*APW+j), +16); B piece j rows [16*(w*BPW+j), +16).
    const int lrow = lane >> 2;                              // 0..15 row inside the piece
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);           // source chunk for LDS position (lane & 3)
    int a_img[APW], a_vy0[APW], a_vx0[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        const int m = m0 + 16 * (wave * APW + i) + lrow;
        if (m < g.M) {
            const int img = m / g.HoWo;
            const int rem = m - img * g.HoWo;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_img[i] = img;
            a_vy0[i] = oy * p.stride - p.pad_top;
            a_vx0[i] = ox * p.stride - p.pad_left;
        } else {
            a_img[i] = -1; a_vy0[i] = 0; a_vx0[i] = 0;
        }
    }
    // split-K: blockIdx.y owns K-steps [ks0, ks0 + nk) and writes an fp32 partial tile to the workspace
    const int ks0 = blockIdx.y * g.nk;
    const int nk = (g.nk_total - ks0) < g.nk ? (g.nk_total - ks0) : g.nk;
    const unsigned char* b_ptr[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i)
        b_ptr[i] = reinterpret_cast<const unsigned char*>(wt + (int64_t)(n0 + 16 * (wave * BPW + i) + lrow) * p.K_pad + kc * 8 + (int64_t)ks0 * BK);
    const unsigned char* a_ptr[APW];
    int a_inc[APW];
    const int steps_per_tap = p.Cin / BK;
    const int ild = p.in_ld > 0 ? p.in_ld : p.Cin;   // physical channels per pixel row
    const int wrap_at = ild / BK;                    // K-steps of a tap at / past this one re-read the row from its start (w_lo segment)
    int kin = ks0 % steps_per_tap;           // wave-uniform cursor of the NEXT step to issue
    int tap_r = (ks0 / steps_per_tap) / p.S, tap_s = (ks0 / steps_per_tap) % p.S;
    bool fresh = true;                       // the first issue of a split may start in the middle of a tap

    auto issue = [&](int stage) {
        if (kin == 0 || fresh) {
#pragma unroll
            for (int i = 0; i < APW; ++i) {
                const int vy = a_vy0[i] + tap_r, vx = a_vx0[i] + tap_s;
                const bool ok = a_img[i] >= 0 && (unsigned)vy < (unsigned)g.Hv && (unsigned)vx < (unsigned)g.Wv;
                const int iy = vy >> p.upsample, ix = vx >> p.upsample;
                const int64_t pix = ((int64_t)a_img[i] * p.H + iy) * p.W + ix;
                a_ptr[i] = ok ? reinterpret_cast<const unsigned char*>(in + pix * ild + kc * 8 + (kin >= wrap_at ? kin - wrap_at : kin) * BK)
                              : reinterpret_cast<const unsigned char*>(g_zero_page);
                a_inc[i] = ok ? BK * 2 : 0;
            }
            fresh = false;
        } else if (kin == wrap_at) {
#pragma unroll
            for (int i = 0; i < APW; ++i) a_ptr[i] -= (int64_t)a_inc[i] * wrap_at;
        }
        const unsigned sa = lds_base + stage * STAGE_BYTES + (16 * wave * APW) * 64;
        const unsigned sb = lds_base + stage * STAGE_BYTES + A_BYTES + (16 * wave * BPW) * 64;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            glds16(a_ptr[i], __builtin_amdgcn_readfirstlane(sa + i * 1024));
            a_ptr[i] += a_inc[i];
        }
#pragma unroll
        for (int i = 0; i < BPW; ++i) {
            glds16(b_ptr[i], __builtin_amdgcn_readfirstlane(sb + i * 1024));
            b_ptr[i] += BK * 2;
        }
        if (++kin == steps_per_tap) { kin = 0; if (++tap_s == p.S) { tap_s = 0; ++tap_r; } }
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if constexpr (ABL != 1) { issue(0); if (nk > 1) issue(1); }

    // fragment read offsets (swizzled): row = base + (lane & 31); chunk = 2*ks + (lane >> 5)
    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    const int foff0 = frow * 64 + (((lane >> 5)) ^ fsw) * 16;        // ks = 0
    const int foff1 = frow * 64 + ((2 + (lane >> 5)) ^ fsw) * 16;    // ks = 1

    // K loop unrolled by the ring depth: every LDS address (fragment reads, DMA destinations) is then a
    // per-lane base + compile-time immediate; what is left per K-step is 3 pointer increments, the DMA
    // issues, 12 ds_read_b128 and 16 MFMAs (the dynamic-stage version spent ~4 VALU + 5 SALU per MFMA).
    const unsigned char* fa0 = lds + (wm * WTM) * 64 + foff0;
    const unsigned char* fa1 = lds + (wm * WTM) * 64 + foff1;
    const unsigned char* fb0 = lds + A_BYTES + (wn * WTN) * 64 + foff0;
    const unsigned char* fb1 = lds + A_BYTES + (wn * WTN) * 64 + foff1;
    auto kstep = [&](auto stage_c, const int kt) {
        constexpr int stage = decltype(stage_c)::value;
        if constexpr (ABL != 1) {
            if (kt + 1 < nk) wait_vmcnt<PIECES>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr bool LATE = ABL == 0;        // LDS-DMA issue after the step's first half of MFMAs: +3...5 % on the Flux linears, S step +-0 (ABL 5 = issue in front, OMGSR_DMA_VARIANT=0, for A/B)
        if constexpr (ABL != 1 && !LATE) {
            if (kt + 2 < nk) issue((stage + 2) % NSTAGE);
        }
        if constexpr (ABL == 3) return;
        x8_t<T> af[2][FM], bf[2][FN];
        if constexpr (ABL == 6) {              // no fragment reads: MFMAs on whatever the registers hold
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < FM; ++i) asm volatile("" : "=v"(af[ks][i]));
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" : "=v"(bf[ks][j]));
            }
        } else {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                af[0][i] = *reinterpret_cast<const x8_t<T>*>(fa0 + stage * STAGE_BYTES + i * 32 * 64);
                af[1][i] = *reinterpret_cast<const x8_t<T>*>(fa1 + stage * STAGE_BYTES + i * 32 * 64);
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bf[0][j] = *reinterpret_cast<const x8_t<T>*>(fb0 + stage * STAGE_BYTES + j * 32 * 64);
                bf[1][j] = *reinterpret_cast<const x8_t<T>*>(fb1 + stage * STAGE_BYTES + j * 32 * 64);
            }
        }
        if constexpr (ABL == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < FM; ++i) asm volatile("" :: "v"(af[ks][i]));
#pragma unroll
                for (int j = 0; j < FN; ++j) asm volatile("" :: "v"(bf[ks][j]));
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mfma32(bf[ks][j], af[ks][i], acc[i][j]);   // transposed tile
                if constexpr (LATE) {
                    if (ks == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (kt + 2 < nk) issue((stage + 2) % NSTAGE);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };
    for (int kt = 0; kt < nk; kt += NSTAGE) {
        kstep(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nk) kstep(std::integral_constant<int, 1>{}, kt + 1);
        if (kt + 2 < nk) kstep(std::integral_constant<int, 2>{}, kt + 2);
    }

    if constexpr (ABL == 4) { if (p.alpha != 12345.0f) return; }      // no epilogue (never true at run time)
    static_assert(NW * 32 * (WTN + 4) * 4 <= lds_bytes(BM, WGN), "epilogue staging must fit the allocation");
    float* epi = reinterpret_cast<float*>(lds) + wave * 32 * (WTN + 4);
    if (g.splits > 1) {
        // raw fp32 partial [split][M][ntn*BN]; bias / activation / residual run in the split-K reduce kernel
        omgsr_igemm_args q = p;
        const int ldw = g.ntn * BN;
        q.out = (float*)p.workspace + (int64_t)blockIdx.y * g.M * ldw;
        q.out_dtype = OMGSR_OUT_F32; q.out_layout = OMGSR_LAYOUT_NHWC; q.out_ld = ldw; q.Cout = ldw;
        q.bias = nullptr; q.gate = nullptr; q.residual = nullptr; q.act = OMGSR_ACT_NONE; q.alpha = 1.0f; q.gn_partial = nullptr; q.out_lo_off = 0;
        q.out_mx = 0; q.overflow_flag = nullptr;           // (round 5: the reduce pass writes the MX form; a partial is plain fp32)
        igemm_epilogue_linear<T, WTN, FM, FN>(q, g.M, acc, epi, lane, m0 + wm * WTM, n0 + wn * WTN, 0);
        return;
    }
    igemm_epilogue_linear<T, WTN, FM, FN>(p, g.M, acc, epi, lane, m0 + wm * WTM, n0 + wn * WTN, bz, p.gn_partial ? g.HoWo : 0);
}

template <int WGM, int WGN, int ABL = 0, int BM = 256>
int launch_dma(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    g.ntm = (g.M + BM - 1) / BM;
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    g.ntn = (logical_cols + WGN * WTN - 1) / (WGN * WTN);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_dma_kernel<bf16_t, WGM, WGN, ABL, BM>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(BM, WGN));
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_dma_kernel<f16_t, WGM, WGN, ABL, BM>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes(BM, WGN));
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(g.ntm * g.ntn, g.splits, a.batch);
    OMGSR_DISPATCH_T(hipLaunchKernelGGL((igemm_dma_kernel<T, WGM, WGN, ABL, BM>), grid, dim3(WGM * WGN * 64), lds_bytes(BM, WGN), st, a, g));
    return (int)hipGetLastError();
}

}  // namespace

namespace omgsr {
int igemm_dma_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    g.nk_total = a.K_pad / BK;
    if (g.splits > 1) {                        // split-K always runs the 256 x 128 tile (see igemm_splitk_plan)
        g.nk = (g.nk_total + g.splits - 1) / g.splits;
        return launch_dma<2, 2>(a, g, st);
    }
    g.splits = 1;
    g.nk = g.nk_total;
#ifdef OMGSR_BUILD_ABLATIONS      // python -m omgsr_amd.build with OMGSR_BUILD_ABLATIONS=1: the experiment-only instantiations (profiles/r02_dma_ablation.md);
                                  // the shipped library leaves them out (20 kernels, 2/3 of this file's compile time)
    static const char* shape = getenv("OMGSR_DMA_WAVES");      // A/B runs: "256" = allow the lockstep 8-wave 256 x 256 tile again
    static const char* abl = ablation_env("OMGSR_DMA_ABLATE");      // timing experiments only: results are garbage (needs OMGSR_ABLATION_OK=1)
    static const char* var = getenv("OMGSR_DMA_VARIANT");     // A/B runs: "0" = LDS-DMA issued in front of the MFMAs
    const bool early = var && var[0] == '0';
    if (abl && abl[0] == '1') return launch_dma<2, 2, 1>(a, g, st);
    if (abl && abl[0] == '2') return launch_dma<2, 2, 2>(a, g, st);
    if (abl && abl[0] == '3') return launch_dma<2, 2, 3>(a, g, st);
    if (abl && abl[0] == '4') return launch_dma<2, 2, 4>(a, g, st);
    if (abl && abl[0] == '6') return launch_dma<2, 4, 6>(a, g, st);      // 256 x 256 tile: no fragment reads
    if (abl && abl[0] == '7') return launch_dma<2, 4, 1>(a, g, st);      // ... no DMA
    if (abl && abl[0] == '8') return launch_dma<2, 4, 4>(a, g, st);      // ... no epilogue
    if (abl && abl[0] == '9') return launch_dma<2, 4, 3>(a, g, st);      // ... DMA only
    // 256 x 256 tile (8 waves of 128x64): a third fewer operand bytes per FLOP; needs 256-row weight padding and
    // enough tiles to fill the chip
    const int64_t t256 = (int64_t)g.ntm * ((logical_cols + 255) / 256) * a.batch;
    // ... and no more padded columns than the 128-wide grid would compute (N = 320 is 512 columns of 256-wide tiles but 384
    // of 128-wide ones: q/k/v/out of the UNet's first level 108 -> 89 us)
    const int cols256 = ((logical_cols + 255) / 256) * 256, cols128 = ((logical_cols + 127) / 128) * 128;
    // (since the ping-pong kernel took the long-K GEMMs the lockstep tile loses to two independent 256 x 128 workgroups per CU on
    // everything that is left: short K, epilogue-bound - S-1024 step -0.7 ms, F-1024 -1.9 ms without it; kept for A/B only)
    if ((shape && shape[0] == '2') && (a.Cout_pad % 256) == 0 && logical_cols >= 256 && t256 >= 200 &&
        cols256 * 16 <= cols128 * 17)
        return early ? launch_dma<2, 4, 5>(a, g, st) : launch_dma<2, 4>(a, g, st);
#else
    const bool early = false;
#endif
    // tile-count quantisation on the 512 slots (2 workgroups per CU): when the 256-row grid leaves the last round under
    // ~3/4 full and the 192-row grid fills it better, take 192 x 128
    {
        const int64_t nt = (logical_cols + 127) / 128;
        const int64_t t256b = (int64_t)((g.M + 255) / 256) * nt * a.batch, t192 = (int64_t)((g.M + 191) / 192) * nt * a.batch;
        // Small-M regime (round 5: one image per call - GEMM-shaped problems of 64 ... 4096 rows that used to fall to the register-staged kernel): a few
        // 256-row tiles would leave most CUs idle and make every K-step 16 MFMAs long on one wave per SIMD; 64 / 128-row tiles give 4x / 2x the
        // workgroups and 4 / 8 MFMAs per step (igemm.hip sends such problems here only when they are GEMM-shaped)
        static const char* bms = getenv("OMGSR_DMA_BM");             // A/B runs: "256" | "192" | "128" | "64" (the last two: GEMM-shaped problems only)
        const bool gemm_shaped = a.R == 1 && a.S == 1 && a.stride == 1 && !a.upsample;
        if (!bms) {
            // GEMM-shaped problems only: the 64 / 128-row instantiations are tested on 1x1 shapes; a 3x3 / strided conv forced here
            // (OMGSR_IGEMM_MODE=dma) keeps the 256 / 192-row tiles whatever its tile count
            if (gemm_shaped && t256b < 64) return launch_dma<2, 2, 0, 64>(a, g, st);
            if (gemm_shaped && t256b < 160) return launch_dma<2, 2, 0, 128>(a, g, st);
        } else if (gemm_shaped) {
            static const char* mk = getenv("OMGSR_DMA_BM_MAXK");      // ... with K_pad <= this (default: any)
            static const int maxk = mk ? atoi(mk) : (1 << 30);
            if (!strcmp(bms, "64") && a.K_pad <= maxk) return launch_dma<2, 2, 0, 64>(a, g, st);
            if (!strcmp(bms, "128") && a.K_pad <= maxk) return launch_dma<2, 2, 0, 128>(a, g, st);
        }
        auto eff = [](int64_t tiles) { const int64_t rounds = (tiles + 511) / 512; return (double)tiles / (double)(rounds * 512); };
        const char* bm = bms;
        const bool force192 = bm && !strcmp(bm, "192"), force256 = bm && !strcmp(bm, "256");
        if (!force256 && (g.M % 192) == 0 && (force192 || (eff(t256b) < 0.78 && eff(t192) > eff(t256b) + 0.1)))
            return launch_dma<2, 2, 0, 192>(a, g, st);
    }
#ifdef OMGSR_BUILD_ABLATIONS
    if (early) return launch_dma<2, 2, 5>(a, g, st);
#endif
    (void)early;
    return launch_dma<2, 2>(a, g, st);
}
}  // namespace omgsr
