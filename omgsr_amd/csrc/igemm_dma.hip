// Implicit-GEMM conv / linear, LDS-DMA variant for large problems (gfx950).
//
// Why a second kernel: the register-staged kernel (igemm.hip) pays one ds_write_b128 per staged
// 16 bytes, and gfx950's register->LDS path sustains only ~79 B/clk/CU (MI355X_MICROARCH.md §LDS):
// at a 128x128 tile that is ~0.8 LDS-write cycles per MFMA cycle, which caps the kernel near 30 % of
// the MFMA peak. Here the operands go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave
// instruction, no VGPR round trip, no ds_write), the tile is 256 x 128 (8 waves, 64x64 per wave:
// 12 KB staged per MFLOP instead of 16) and three K-stages are in flight.
//
//  * LDS image is LINEAR ([row][64 B], BK = 32) because the DMA writes base + lane*16; bank
//    conflicts are removed by a swizzle applied to the SOURCE chunk and to the read
//    (position = chunk ^ ((row >> 2) & 3)): every 16-lane ds_read_b128 service group then covers
//    16 distinct 16-B slots
//  * the conv halo / rows past M / taps past R*S read a 16-byte ZERO PAGE instead of being predicated
//    (an LDS-DMA lane that is masked off would leave stale bytes in its slot)
//  * 3-deep LDS ring, prefetch distance 2, ONE raw s_barrier per K-step; waits are counted
//    (s_waitcnt vmcnt(3) = "my copies for this step have landed, next step's may still fly");
//    the DMA is issued from inline asm so hipcc neither drains it at the barrier nor before ds_reads
//  * epilogue shared with igemm.hip
#include "common.hip.h"
#include "../../include/omgsr_hip.h"
#include "igemm_epilogue.hip.h"

namespace {

constexpr int BK = 32;
constexpr int NT = 512;            // 8 waves
constexpr int BM = 256, BN = 128;
constexpr int WGM = 4, WGN = 2;
constexpr int WTM = 64, WTN = 64, FM = 2, FN = 2;
constexpr int A_BYTES = BM * BK * 2;          // 16 KB
constexpr int B_BYTES = BN * BK * 2;          //  8 KB
constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
constexpr int NSTAGE = 3;
constexpr int EPI_LD = WTN + 4;
constexpr int EPI_BYTES = 8 * 32 * EPI_LD * 4;
constexpr int LDS_BYTES = (NSTAGE * STAGE_BYTES > EPI_BYTES) ? NSTAGE * STAGE_BYTES : EPI_BYTES;

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

__global__ __launch_bounds__(NT) void igemm_dma_kernel(const omgsr_igemm_args p, const IgemmGeo g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // LDS_BYTES (72 KB), dynamic

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WGN, wn = wave % WGN;

    const int tile = xcd_remap(blockIdx.x, g.ntm * g.ntn);
    const int tn = tile % g.ntn, tm = tile / g.ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.z;

    const bf16_t* __restrict__ in = (const bf16_t*)p.in + (int64_t)bz * p.in_bstride;
    const bf16_t* __restrict__ wt = (const bf16_t*)p.weight + (int64_t)bz * p.w_bstride;
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    const unsigned lds_base = (unsigned)(size_t)(lds_byte_t*)lds;            // LDS byte offset of the ring

    // ---- DMA coordinates: a wave instruction fills 16 rows x 64 B ---------------------------
    // A: wave w fills rows [32w, 32w+32) with two pieces; B: rows [16w, 16w+16) with one piece.
    const int lrow = lane >> 2;                              // 0..15 row inside the piece
    const int kc = (lane & 3) ^ ((lane >> 4) & 3);           // source chunk for LDS position (lane & 3)
    int a_img[2], a_vy0[2], a_vx0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + 32 * wave + 16 * i + lrow;
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
    // K order is tap-major: K-step kt = (tap, channel step). Cin % 32 == 0 here, so a step never
    // straddles taps: the (validity, pointer) pair of each row is recomputed once per TAP and the
    // steps inside a tap only advance the pointers by 64 B (0 B for rows parked on the zero page):
    // ~3 VALU per K-step instead of ~80 (the kernel was instruction-issue bound, profiles/r01_pmc_*).
    const unsigned char* b_ptr = reinterpret_cast<const unsigned char*>(wt + (int64_t)(n0 + 16 * wave + lrow) * p.K_pad + kc * 8);
    const unsigned char* a_ptr[2];
    int a_inc[2];
    const int steps_per_tap = p.Cin / BK;
    int kin = 0, tap_r = 0, tap_s = 0;       // wave-uniform cursor of the NEXT step to issue

    auto issue = [&](int stage) {
        if (kin == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int vy = a_vy0[i] + tap_r, vx = a_vx0[i] + tap_s;
                const bool ok = a_img[i] >= 0 && (unsigned)vy < (unsigned)g.Hv && (unsigned)vx < (unsigned)g.Wv;
                const int iy = vy >> p.upsample, ix = vx >> p.upsample;
                const int64_t pix = ((int64_t)a_img[i] * p.H + iy) * p.W + ix;
                a_ptr[i] = ok ? reinterpret_cast<const unsigned char*>(in + pix * p.Cin + kc * 8)
                              : reinterpret_cast<const unsigned char*>(g_zero_page);
                a_inc[i] = ok ? BK * 2 : 0;
            }
        }
        const unsigned sa = lds_base + stage * STAGE_BYTES + (32 * wave) * 64;
        const unsigned sb = lds_base + stage * STAGE_BYTES + A_BYTES + (16 * wave) * 64;
        glds16(a_ptr[0], __builtin_amdgcn_readfirstlane(sa));
        glds16(a_ptr[1], __builtin_amdgcn_readfirstlane(sa + 1024));
        glds16(b_ptr, __builtin_amdgcn_readfirstlane(sb));
        a_ptr[0] += a_inc[0];
        a_ptr[1] += a_inc[1];
        b_ptr += BK * 2;
        if (++kin == steps_per_tap) { kin = 0; if (++tap_s == p.S) { tap_s = 0; ++tap_r; } }
    };

    f32x16_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    issue(0);
    if (g.nk > 1) issue(1);

    // fragment read offsets (swizzled): row = base + (lane & 31); chunk = 2*ks + (lane >> 5)
    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    const int foff0 = frow * 64 + (((lane >> 5)) ^ fsw) * 16;        // ks = 0
    const int foff1 = frow * 64 + ((2 + (lane >> 5)) ^ fsw) * 16;    // ks = 1

    int stage = 0;
    for (int kt = 0; kt < g.nk; ++kt) {
        if (kt + 1 < g.nk) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < g.nk) {
            int s2 = stage + 2; if (s2 >= NSTAGE) s2 -= NSTAGE;
            issue(s2);
        }
        const unsigned char* As = lds + stage * STAGE_BYTES + (wm * WTM) * 64;
        const unsigned char* Bs = lds + stage * STAGE_BYTES + A_BYTES + (wn * WTN) * 64;
        bf16x8_t af[2][FM], bf[2][FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            af[0][i] = *reinterpret_cast<const bf16x8_t*>(As + i * 32 * 64 + foff0);
            af[1][i] = *reinterpret_cast<const bf16x8_t*>(As + i * 32 * 64 + foff1);
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            bf[0][j] = *reinterpret_cast<const bf16x8_t*>(Bs + j * 32 * 64 + foff0);
            bf[1][j] = *reinterpret_cast<const bf16x8_t*>(Bs + j * 32 * 64 + foff1);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mfma32(af[ks][i], bf[ks][j], acc[i][j]);
        if (++stage == NSTAGE) stage = 0;
    }

    float* epi = reinterpret_cast<float*>(lds) + wave * 32 * EPI_LD;
    igemm_epilogue<WTN, FM, FN>(p, g.M, acc, epi, lane, m0 + wm * WTM, n0 + wn * WTN, bz);
}

}  // namespace

namespace omgsr {
int igemm_dma_launch(const omgsr_igemm_args& a, IgemmGeo g, hipStream_t st) {
    const int logical_cols = (a.act == OMGSR_ACT_GEGLU) ? 2 * a.Cout : a.Cout;
    g.nk = a.K_pad / BK;
    g.ntm = (g.M + BM - 1) / BM;
    g.ntn = (logical_cols + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_dma_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    dim3 grid(g.ntm * g.ntn, 1, a.batch);
    hipLaunchKernelGGL(igemm_dma_kernel, grid, dim3(NT), LDS_BYTES, st, a, g);
    return (int)hipGetLastError();
}
}  // namespace omgsr
